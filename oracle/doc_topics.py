"""TEST INFRASTRUCTURE ONLY (oracle): the per-entity topic proportions of printDocumentTopics
(src/main/java/org/madgik/MVTopicModel/FastQMVWVParallelTopicModel.java:2871-2899; the same lines in
FastQMVWVTopicInferencer.java:383-411) and the text it prints (:2862-2869, :2902-2909), in numpy / plain Python.
Only tests/ may import this file.  Parity unpinned by the reference (no tests, no JVM); the arithmetic order is
the source's: ((w * (n_dk + gamma*alpha)) / (len + gamma*alphaSum)) summed over the views, divided by sum w."""
import numpy as np


def doc_topic_proportions(K, doc_off, z, alpha, alpha_sum, gamma, w):
    """doc_off[m]: int64[D+1]; z[m]: int32[N_m]; alpha: [M][K+1]; w[m] = (m==0 ? 1 : discr[m]) * pMean[0][m]."""
    M = len(doc_off)
    D = len(doc_off[0]) - 1
    out = np.zeros((D, K), dtype=np.float64)
    norm = np.float64(0.0)
    for m in range(M):
        norm = norm + np.float64(w[m])
    # topicCounts[m] / docLen[m] live outside the entity loop and are refreshed only when the entity has view m
    # (FastQMVWVParallelTopicModel.java:2873-2886): a missing view is scored with the previous holder's values
    cnt = [np.zeros(K, dtype=np.float64) for _ in range(M)]
    doc_len = [0] * M
    for d in range(D):
        tp = np.zeros(K, dtype=np.float64)
        for m in range(M):
            b, e = int(doc_off[m][d]), int(doc_off[m][d + 1])
            if e > b:                                   # Assignments[m] != null (an empty span stands for null)
                zz = z[m][b:e]
                cnt[m] = np.bincount(zz[zz >= 0], minlength=K).astype(np.float64)
                doc_len[m] = e - b
            num = cnt[m] + np.float64(gamma[m]) * np.asarray(alpha[m][:K], dtype=np.float64)
            den = np.float64(doc_len[m]) + np.float64(gamma[m]) * np.float64(alpha_sum[m])
            tp = tp + (np.float64(w[m]) * num) / den
        out[d] = tp / norm
    return out


def print_document_topics(prop, names, threshold, max_topics, fmt):
    """The reference's text: the builder grows by 'topic<TAB>weight<TAB>' and is printed after every addition."""
    D, K = prop.shape
    if max_topics < 0 or max_topics > K:
        max_topics = K
    lines = ["#doc name topic proportion ..."]
    for d in range(D):
        # cc.mallet.types.IDSorter.compareTo (mallet-2.0.8 class file): weight descending, equal weights by id DESCENDING
        order = sorted(range(K), key=lambda k: (-prop[d, k], -k))
        builder = f"{d}\t{names[d]}\t"
        for i in range(max_topics):
            k = order[i]
            if prop[d, k] < threshold:
                break
            builder += f"{k}\t{fmt(prop[d, k])}\t"
            lines.append(builder)
    return "\n".join(lines) + "\n"
