"""Numerics of an "emulated fp32" convolution stack on the bf16 matrix cores (CPU only, no kernel):
every 3x3 / 1x1 / 7x7 convolution and linear layer of the oracle's ResNet-18 + head with each fp32 operand split into bf16
slices (x = x0 + x1 + x2, each slice the bf16 rounding of what is left) and the product taken as the sum of the leading
cross terms, accumulated in fp32 -- what `v_mfma_f32_32x32x16_bf16` would compute (bf16 x bf16 products are exact in fp32).
Reports max |dlogp| against a float64 run of the same operator for: plain fp32, 6 terms (i + j <= 2), 3 terms (i + j <= 1),
1 term (plain bf16 operands, fp32 accumulate).  Checker-side study (it lives under tests/ because it imports oracle/): never shipped, not collected by pytest.
  python tests/emulated_fp32_study.py [windows]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cnn  # noqa: E402
from playaid_core_amd import synth  # noqa: E402


def slices(t, n):
    out, rest = [], t
    for _ in range(n):
        s = rest.to(torch.bfloat16).to(torch.float32)
        out.append(s)
        rest = rest - s
    return out


def emulated(op, terms):
    def f(x, w, *a, **k):
        if x.dtype != torch.float32:
            return op(x, w, *a, **k)
        bias = a[0] if a else k.get("bias")
        a2 = (None,) + tuple(a[1:]) if a else a
        k2 = dict(k)
        if "bias" in k2:
            k2["bias"] = None
        n = {1: 1, 3: 2, 6: 3}[terms]
        xs, ws = slices(x, n), slices(w, n)
        acc = None
        for i in range(n):
            for j in range(n):
                if i + j > n - 1:
                    continue
                y = op(xs[i], ws[j], *a2, **k2)
                acc = y if acc is None else acc + y
        if bias is not None:
            acc = acc + (bias.view(1, -1, 1, 1) if acc.dim() == 4 else (bias.view(1, -1, 1) if acc.dim() == 3 else bias))
        return acc
    return f


def main():
    n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    sd = synth.make_state_dict()
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.random((n_win, 7, 3, 128, 128), dtype=np.float32))
    ref64 = cnn.forward(x.double(), sd).numpy()
    plain = cnn.forward(x, sd).numpy()
    print(f"{n_win} windows x 7 crops, ResNet-18 + Conv1d + MLP head, max |dlogp| against the float64 run (bar of the path: 1e-4)")
    print(f"  plain fp32                         {np.abs(plain - ref64).max():.3e}")
    conv2d, conv1d, linear = F.conv2d, F.conv1d, F.linear
    for terms in (6, 3, 1):
        F.conv2d, F.conv1d, F.linear = emulated(conv2d, terms), emulated(conv1d, terms), emulated(linear, terms)
        try:
            got = cnn.forward(x, sd).numpy()
        finally:
            F.conv2d, F.conv1d, F.linear = conv2d, conv1d, linear
        name = {6: "6 cross terms (3 slices, i + j <= 2)", 3: "3 cross terms (2 slices, i + j <= 1)", 1: "1 term (plain bf16 operands)     "}[terms]
        print(f"  {name}  {np.abs(got - ref64).max():.3e}   (against plain fp32: {np.abs(got - plain).max():.3e})")
    print("matrix-pipe cost per fp32 multiply-add: fp32 MFMA 1 (32x32x2 every 64 cycles); 6 bf16 terms 6/16; 3 terms 3/16; 1 term 1/16")


if __name__ == "__main__":
    main()
