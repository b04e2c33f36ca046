"""KAT-7: an oracle-independent known answer for the MULTI-VIEW conditional of WRK:301-601 at a mid-document
position, written from the formulas of SURVEY.md §8a step 4 (WRK:399-418, 434-468, 501-519; PTM:2670-2678) in
numpy.  Nothing here calls into oracle/ or the HIP library for the expected value: the only thing taken from a
sampler run is the HISTORY (which topics the earlier tokens of the entity were given), which is an input of the
conditional, not part of its arithmetic.

The scenario is hand-made: 3 views, K = 7, uneven alpha / beta / gamma, an explicit view-weight matrix p, global
counts n_wk / n_k that are plain numbers (the sampler only reads them), one entity whose views share some topics.
`expected_conditional` replays the dense-list bookkeeping (topics present in any view at entry; a topic leaves the
list when its count reaches zero in every view, WRK:441-468; the list never grows, Q1/Q2; the other-view mass is
frozen per view and is zero outside the list, Q3) and returns the normalised K+1 vector
    P(k) ~ 1[k in list]*(p_mm*n_dk[m][k] + other_k)*p_wt(k) + leaf_k,   slot K = newTopicMass.
"""
import numpy as np

K = 7
V = [12, 5, 6]
M = 3

# entity 0 is the one under test; entity 1 is a bystander (different views present)
TOKENS = [
    [[3, 7, 7, 1, 11, 0], [2, 2, 9]],          # view 0
    [[4, 0, 2, 2, 1], []],                     # view 1 (entity 1 lacks it: Assignments[1] == null, MTA:19)
    [[5, 5, 3], [0, 1]],                       # view 2
]
Z0 = [
    [[2, 5, 5, 0, 2, 6], [1, 1, 3]],
    [[4, 5, 2, 2, 0], []],                     # topic 4 lives in this view only, once: its decrement removes it
    [[5, 6, 2], [3, 3]],
]


def corpus():
    doc_off, toks, z = [], [], []
    for m in range(M):
        lens = [len(t) for t in TOKENS[m]]
        doc_off.append(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
        toks.append(np.asarray([w for d in TOKENS[m] for w in d], dtype=np.int32))
        z.append(np.asarray([t for d in Z0[m] for t in d], dtype=np.int32))
    return doc_off, toks, z


def hyper(with_inactive):
    rng = np.random.RandomState(2024)
    alpha = 0.05 + rng.rand(M, K + 1) * 0.4                  # alpha[m][K] = new-topic weight PTM:196
    alpha_sum = alpha[:, :K].sum(axis=1) * np.array([1.0, 0.9, 1.1])   # an input of its own, not re-derived by the sampler
    beta = np.array([0.013, 0.021, 0.008])
    beta_sum = beta * np.asarray(V, dtype=np.float64)
    gamma = np.array([1.0, 0.7, 1.3])
    inactive = None
    if with_inactive:
        inactive = np.zeros(K, dtype=np.uint8)
        inactive[[1, 3]] = 1                                 # topics no token of entity 0 holds; first inactive = 1
    return dict(alpha=alpha, alpha_sum=alpha_sum, beta=beta, beta_sum=beta_sum, gamma=gamma, inactive=inactive)


def view_weights():
    """p[D][M][M] as WRK:327-337 leaves it: symmetric, unit diagonal, three decimals."""
    p = np.zeros((2, M, M))
    p[0] = [[1.0, 0.412, 0.087], [0.412, 1.0, 0.650], [0.087, 0.650, 1.0]]
    p[1] = [[1.0, 0.5, 0.25], [0.5, 1.0, 0.125], [0.25, 0.125, 1.0]]
    return p


def global_counts():
    rng = np.random.RandomState(7)
    nwk = [rng.randint(0, 9, size=(V[m], K)).astype(np.int32) for m in range(M)]
    nk = [(nwk[m].sum(axis=0) + rng.randint(0, 5, size=K)).astype(np.int32) for m in range(M)]
    return nwk, nk


def expected_conditional(hy, p, nwk, nk, z_before, z_after, m_t, pos_t, d=0):
    """Normalised K+1 conditional of token (entity d, view m_t, position pos_t), given the assignments at entry
    (z_before[m] = list per view for the entity) and what the earlier tokens were resampled to (z_after).
    Also returns the bookkeeping facts the test asserts on."""
    alpha, alpha_sum, beta, beta_sum, gamma = hy["alpha"], hy["alpha_sum"], hy["beta"], hy["beta_sum"], hy["gamma"]
    inactive = hy["inactive"] if hy["inactive"] is not None else np.zeros(K, dtype=np.uint8)
    first_inactive = int(np.flatnonzero(inactive)[0]) if inactive.any() else -1
    lens = [len(z_before[m]) for m in range(M)]
    ndk = np.zeros((M, K))
    for m in range(M):
        for t in z_before[m]:
            ndk[m, t] += 1                                                   # WRK:339-361
    in_list = ndk.sum(axis=0) > 0                                            # WRK:376-391
    facts = dict(removed=[], entered_outside_list=[], reentered_removed=[])
    for m in range(M):                                                       # WRK:393
        if lens[m] == 0:
            continue
        scale = lens[m] + gamma[m] * alpha_sum[m]
        other = np.zeros(K)
        for k in range(K):                                                   # WRK:399-410, frozen for this view (Q3)
            if not in_list[k]:
                continue
            acc = 0.0
            for i in range(M):
                if i != m and lens[i] != 0:
                    acc += p[d][m][i] * (ndk[i, k] + gamma[i] * alpha[i, k]) / (lens[i] + gamma[i] * alpha_sum[i])
            other[k] = acc * scale
        new_all = 0.0
        for i in range(M):                                                   # WRK:413-418 (i == m and empty views included)
            new_all += p[d][m][i] * (gamma[i] * alpha[i, K]) / (lens[i] + gamma[i] * alpha_sum[i])
        new_all *= scale
        for pos in range(lens[m]):
            old = z_before[m][pos]
            ndk[m, old] -= 1                                                 # WRK:434-437
            if ndk[:, old].sum() == 0 and in_list[old]:                      # WRK:441-468
                in_list[old] = False
                facts["removed"].append((m, pos, old))
            if m == m_t and pos == pos_t:
                w = TOKENS[m][d][pos]
                p_wt = (nwk[m][w] + beta[m]) / (nk[m] + beta_sum[m])         # WRK:507, own count included (Q5)
                doc_term = np.where(in_list, (p[d][m][m] * ndk[m] + other) * p_wt, 0.0)      # WRK:509
                leaf = np.where(inactive != 0, 0.0, gamma[m] * alpha[m, :K] * p_wt)         # PTM:2670-2678
                new_mass = 0.0 if first_inactive < 0 else new_all / K                        # WRK:515 (Q7)
                total = new_mass + doc_term.sum() + leaf.sum()                               # WRK:519
                out = np.zeros(K + 1)
                out[:K] = (doc_term + leaf) / total
                out[K] = new_mass / total
                facts["in_list"] = in_list.copy()
                facts["ndk"] = ndk.copy()
                return out, facts
            new = z_after[m][pos]
            if not in_list[new]:
                (facts["reentered_removed"] if any(r[2] == new for r in facts["removed"]) else facts["entered_outside_list"]).append((m, pos, new))
            ndk[m, new] += 1                                                 # WRK:557-560; the list does not grow (Q1/Q2)
    raise ValueError("token not found")


def entity_slices(doc_off, z, d=0):
    return [list(map(int, z[m][doc_off[m][d]:doc_off[m][d + 1]])) for m in range(M)]
