/* mvhdp_oracle_internal.h — CPU ORACLE internals (test infrastructure, NOT product code). */
#ifndef MVHDP_ORACLE_INTERNAL_H
#define MVHDP_ORACLE_INTERNAL_H
#include "mvhdp_oracle.h"

struct orc_model {
    int K, M;
    int32_t V[ORC_MAX_M];
    int64_t rowbase[ORC_MAX_M + 1]; /* cumulative V */
    int64_t D;
    int64_t* doc_off[ORC_MAX_M];
    int32_t* tokens[ORC_MAX_M];
    int32_t* z[ORC_MAX_M];
    int64_t N[ORC_MAX_M];
    int32_t* nwk;   /* [sumV][K]  typeTopicCounts */
    int32_t* nk;    /* [M][K]     tokensPerTopic  */
    double*  trees; /* [sumV][2K] FTree.tree      */
    double*  alpha; /* [M][K+1] */
    double alpha_sum[ORC_MAX_M], beta[ORC_MAX_M], beta_sum[ORC_MAX_M], gamma[ORC_MAX_M];
    double p_a[ORC_MAX_M][ORC_MAX_M], p_b[ORC_MAX_M][ORC_MAX_M];
    uint8_t* inactive; /* [K] */
};

#endif
