"""CPU study behind csrc/bwd_exact.hip: why the bias gradients of TINY batches leave the 1e-3 gate under the fp16 backward
arithmetic, and what would and would not fix it.  Re-creates the cases of tests/tools/fuzz_parity.py (seed 1) that miss the gate
-- 23, 37, 57 -- and two that hold it, and evaluates on the CPU (torch, emulating the kernels' roundings):

  A  the kernels' arithmetic: g_raw, cos, dZ of every layer rounded to fp16, db = sum of the ROUNDED dZ        (what shipped in round 3)
  B  the same chain, db summed from the fp32 dH * cos before its rounding                                    
  C  B + cos in fp32 for the bias terms
  D  the whole chain in fp32                                                                                (csrc/bwd_exact.hip)

and the condition number kappa_l = || sqrt(sum_n t_n^2) || / || sum_n t_n || of every bias sum with the model
error = 2^-12 sqrt(2 (L - l) + 1) kappa_l (tests/conftest.py:fp16_chain_bias_bounds).

Result (python tests/tools/bias_conditioning.py, 1 min): B moves the worst tensor of case 37 from 3.4e-2 to 3.0e-2 and of case 23
from 2.0e-3 to 1.3e-3 -- the error is not made by the last sum, it is carried by every fp16 operand of the chain (the out
layer's bias, whose only operand is g_raw, is the one tensor B fixes) -- while measured error / model stays within 0.3 ... 0.9 for
kappa from 0.1 (32640 samples) to 65 (34 samples).  Only D removes it, which is why batches of <= 4096 samples take the fp32
backward (sunerf_hip/ops.py:EXACT_BACKWARD_SAMPLES)."""
import os
import random
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'oracle')); sys.path.insert(0, os.path.join(R, 'tests'))
import sunerf_oracle as orc   # noqa: E402


def f16(x):
    return x.to(torch.float16).to(torch.float32)


def run(case, d, L, n, S, scale):
    params = orc.init_params(d_filter=d, n_layers=L, seed=1000 + case)
    params = [((W * scale) if 0 < i < len(params) - 1 else W, b) for i, (W, b) in enumerate(params)]
    side = int(n ** 0.5) + 1
    o, dd = orc.synthetic_rays(side)
    o, dd = o[:n].contiguous(), dd[:n].contiguous()
    g = torch.Generator().manual_seed(case)
    t = torch.rand(n, 1, generator=g) * 3
    z = orc.stratified_z(o, dd, orc.linspace_t_vals(S), torch.tensor(1.3), torch.tensor(1.0))
    leaves = [(W.clone().requires_grad_(True), b.clone().requires_grad_(True)) for W, b in params]
    ref = orc.render_pass(leaves, o, dd, t, z)
    ref['raw'].retain_grad()
    g_img = torch.randn(n, 1, generator=g)
    (ref['image'] * g_img).sum().backward()
    g_raw = ref['raw'].grad.reshape(-1, 2)
    pts = orc.points_on_rays(o, dd, z)
    x = torch.cat([pts, t[:, None, :].expand(-1, S, -1)], -1).reshape(-1, 4)
    enc = orc.positional_encoding(x)
    Hs, Cs, h = [], [], enc
    for W, b in params[:-1]:
        zz = h @ W.T + b
        Hs.append(torch.sin(zz)); Cs.append(torch.cos(zz)); h = Hs[-1]
    gs = 2.0 ** (4 - torch.frexp(g_raw.abs().max())[1].item())       # the kernels' power-of-two gradient scale
    from conftest import fp16_chain_bias_bounds
    model = fp16_chain_bias_bounds(params, o, dd, t, z, ref['raw'].grad)
    print(f'case {case}: d={d} L={L} rays={n} S={S} hidden x{scale}  ({n * S} samples)')
    for name, sum32, cos32, chain32 in (('A kernels (r3)', False, False, False), ('B fp32 bias sums', True, False, False),
                                        ('C B + fp32 cos', True, True, False), ('D fp32 chain', True, True, True)):
        dz16 = g_raw * gs if chain32 else f16(g_raw * gs)
        gb, gW = [None] * len(params), [None] * len(params)
        gb[-1] = (g_raw * gs).sum(0) if sum32 else dz16.sum(0)
        gW[-1] = dz16.T @ (Hs[-1] if chain32 else f16(Hs[-1]))
        for l in range(len(params) - 2, -1, -1):
            dH = dz16 @ params[l + 1][0]
            dz32 = dH * (Cs[l] if chain32 else f16(Cs[l]))
            dzb = dH * Cs[l] if cos32 else dz32
            dz16 = dz32 if chain32 else f16(dz32)
            gb[l] = dzb.sum(0) if sum32 else dz16.sum(0)
            X = Hs[l - 1] if l > 0 else enc
            gW[l] = dz16.T @ (X if chain32 else f16(X))
        eb = [((gb[l] / gs - b.grad).norm() / b.grad.norm()).item() for l, (_, b) in enumerate(leaves)]
        ew = [((gW[l] / gs - W.grad).norm() / W.grad.norm()).item() for l, (W, _) in enumerate(leaves)]
        print(f'  {name:18s} worst W {max(ew):.1e} | b ' + ' '.join(f'{v:.1e}' for v in eb))
        if name.startswith('A'):
            print('  ' + ' ' * 18 + ' kappa         | b ' + ' '.join(f'{k:7.1f}' for k, _ in model))
            nl = len(params)
            print('  ' + ' ' * 18 + ' error / model | b ' + ' '.join(
                f'{e / (2.0 ** -12 * (2 * (nl - 1 - l) + 1) ** 0.5 * k):7.2f}' for l, (e, (k, _)) in enumerate(zip(eb, model))))


if __name__ == '__main__':
    torch.set_num_threads(8)
    rng = random.Random(1)
    want = {int(v) for v in sys.argv[1:]} or {16, 19, 23, 37, 57}
    for case in range(60):
        d = rng.choice([64, 64, 128, 256, 256, 512])
        L = rng.randint(1, 8) if d < 512 else rng.randint(1, 3)
        n = rng.choice([1, 2, 3, 5, 17, 33, 64, 100, 255, 300])
        S = rng.choice([2, 3, 31, 32, 33, 64, 65, 96, 127, 128, 130, 200])
        scale = rng.choice([1.0, 1.0, 0.25, 2.0])
        rng.choice([0, 1])
        if case in want:
            run(case, d, L, n, S, scale)
