"""Reference-generated pin for the STOCHASTIC sampler: seeded draws of the reference's own ``sample_token``
(/root/reference/src/sopro/sampling.py:24-93, with the policy constants of src/sopro/model.py:289-290: top_k 50, repetition
penalty 1.1) -> empirical token frequencies, stored in ``tests/golden/sampler.npz``.

The RNG stream of ``torch.multinomial`` on the CPU cannot be matched by a device sampler, so the pin is distributional:
``tests/test_oracle_golden.py`` checks ``oracle.sampling_distribution`` against these frequencies on the CPU and
``tests/test_gpu_ops.py`` checks ``ar_sample_kernel``'s draws against them on the GPU (5 sigma of a two-sample comparison).
Every case carries a token history, so the temperature -> repetition-penalty -> softmax -> top-k -> renormalise -> top-p
(shift-by-one) -> renormalise order of the reference is what the frequencies encode.

Runs only in the build container (needs /root/reference).  Usage:  python tests/golden/make_golden_sampler.py
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")

import numpy as np
import torch

N_DRAWS = 20000
V1 = 2049
# (top_p, temperature, top_k, logit scale): the reference's defaults, a sharper setting, top-p off with a small top-k, and
# the anti-loop recovery setting of src/sopro/model.py:226-229
CASES = [(0.9, 1.05, 50, 2.0), (0.6, 0.9, 50, 4.0), (1.0, 1.0, 8, 1.0), (0.85, 1.2, 50, 1.5)]


def main():
    from sopro.sampling import sample_token

    out = {"n_draws": np.int64(N_DRAWS), "cases": np.asarray(CASES, dtype=np.float64)}
    for ci, (top_p, temp, top_k, scale) in enumerate(CASES):
        g = torch.Generator().manual_seed(4100 + ci)
        logits = (torch.randn(V1, generator=g) * scale).float()
        # history: 50+ tokens with repeats, several of them among the head of the distribution (the penalty then moves mass
        # inside the kept set, which is what distinguishes penalty-before-softmax from any other order)
        head = torch.topk(logits, 12).indices.tolist()
        hist = [int(v) for v in torch.randint(0, 2048, (60,), generator=g)]
        for j, tkn in enumerate(head[:6]):
            hist[-(3 + 5 * j)] = tkn  # inside the last 50
        hist[2] = head[7]  # outside the last 50: must NOT be penalised
        torch.manual_seed(9000 + ci)
        x = logits.view(1, 1, V1)
        counts = np.zeros(V1, dtype=np.int64)
        for _ in range(N_DRAWS):
            counts[sample_token(x, hist, top_p=top_p, top_k=top_k, temperature=temp, repetition_penalty=1.1)] += 1
        out[f"logits{ci}"] = logits.numpy()
        out[f"hist{ci}"] = np.asarray(hist, dtype=np.int64)
        out[f"counts{ci}"] = counts
        print(f"case {ci}: top_p {top_p} T {temp} top_k {top_k}: {int((counts > 0).sum())} distinct tokens, head freq {counts.max() / N_DRAWS:.4f}")
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)


if __name__ == "__main__":
    main()
