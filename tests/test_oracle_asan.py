"""The oracle (test infrastructure) under AddressSanitizer on the inputs that once broke it: out-of-alphabet types in
build_counts used to write past the n_wk rows (and, in the last view, past the heap block -- the `munmap_chunk(): invalid
pointer` abort at the end of a round-1 GPU test run).  CPU only; sanitizers are not available for the GPU build."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "mvhdp_oracle.h"
#include <stdio.h>
int main(void) {
    int32_t V[2] = {30, 12};
    orc_model* o = orc_create(7, 2, V);
    int64_t off0[3] = {0, 4, 7}, off1[3] = {0, 2, 3};
    int32_t t0[7] = {1, 2, 33, 4, 5, 31, 6};
    int32_t t1[3] = {0, 14, 3};                    /* 14: beyond the LAST view's alphabet */
    int32_t z0[7] = {0, 1, 2, 3, 4, 5, 6}, z1[3] = {1, 6, 3};
    double alpha[16], as[2] = {0.7, 0.7}, b[2] = {0.01, 0.01}, bs[2] = {0.3, 0.12}, g[2] = {1, 1}, pa[4] = {0.31, 0.31, 0.31, 0.31}, pb[4] = {1, 1, 1, 1};
    for (int i = 0; i < 16; i++) alpha[i] = 0.1;
    orc_set_corpus(o, 0, 2, off0, t0); orc_set_corpus(o, 1, 2, off1, t1);
    orc_set_assignments(o, 0, z0); orc_set_assignments(o, 1, z1);
    orc_set_hyper(o, alpha, as, b, bs, g, pa, pb, 0);
    orc_build_counts(o);
    orc_stats st;
    for (unsigned it = 0; it < 3; it++) if (orc_sweep(o, it, 5, 0, 0, 0, &st, 0, 0, 0, 0, 0, 0, 0, 0)) return 1;
    printf("tokens %lld oov %lld\n", (long long)st.tokens, (long long)st.oov_skipped);
    orc_destroy(o);
    return 0;
}
'''


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_out_of_alphabet_types_under_asan(tmp_path):
    (tmp_path / "t.c").write_text(SRC)
    exe = str(tmp_path / "t")
    cc = ["gcc", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=gnu11", "-ffp-contract=off",
          "-I", os.path.join(ROOT, "oracle"), "-o", exe, str(tmp_path / "t.c"),
          os.path.join(ROOT, "oracle", "mvhdp_oracle.c"), os.path.join(ROOT, "oracle", "ref_threaded.c"), "-lm", "-lpthread"]
    subprocess.check_call(cc)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "tokens 7 oov 3" in r.stdout
