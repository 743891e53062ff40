"""Pipelined backward (csrc/bwd_pipe.hip) against the two-kernel backward on the same stash: per-tensor relative L2 difference,
status word, and HIP-event timings of both.  Usage: python tools/pipe_check.py [n_rays n_samples [reps]] ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

import sunerf_oracle as orc  # noqa: E402  (initial weights / synthetic rays only)
from sunerf_hip import ops  # noqa: E402


def run(n_side, S, reps, d_filter=256, n_layers=8):
    torch.manual_seed(0)
    dev = torch.device('cuda')
    params = orc.init_params(d_filter=d_filter, n_layers=n_layers, seed=3)
    o, d = orc.synthetic_rays(n_side)
    n = o.shape[0]
    t = torch.rand(n, 1) * 5.
    z = orc.stratified_z(o, d, orc.linspace_t_vals(S), torch.tensor(1.3), torch.tensor(1.0))
    Ws = [W.to(dev) for W, _ in params]
    bs = [b.to(dev) for _, b in params]
    packed = ops.PackedMLP(Ws, bs)
    o, d, t, z = o.to(dev), d.to(dev), t.to(dev), z.to(dev)
    g_image = torch.randn(n, device=dev) * 1e-3
    res = {}
    for mode in ('classic', 'pipe'):
        ops._backward_forced = mode
        fwd = ops.emission_render_fwd(packed, o, d, t, z, reg_radius=1.2, training=True)      # (stash in the format this backward reads)
        gW = [torch.full_like(W, float('nan')) for W in Ws]
        gb = [torch.full_like(b, float('nan')) for b in bs]
        ops.emission_render_bwd(packed, o, d, z, fwd['raw'], fwd['stash'], g_image, None, 2e-5, 1.2, gW, gb)
        torch.cuda.synchronize()
        st = ops.pipe_status(raise_on_failure=False) if mode == 'pipe' else 0
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            ops.emission_render_bwd(packed, o, d, z, fwd['raw'], fwd['stash'], g_image, None, 2e-5, 1.2, gW, gb)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / max(reps, 1)
        if mode == 'pipe':
            st = max(st, ops.pipe_status(raise_on_failure=False))
        res[mode] = (gW, gb, ms, st)
    ops._backward_forced = None
    worst = 0.
    lines = []
    for i in range(len(Ws)):
        for name, a, b in (('W', res['classic'][0][i], res['pipe'][0][i]), ('b', res['classic'][1][i], res['pipe'][1][i])):
            e = ((a - b).norm() / a.norm()).item() if torch.isfinite(b).all() else float('nan')
            lines.append(f'{name}{i}:{e:.2e}')
            worst = max(worst, e) if e == e else float('nan')
    print(f'rays {n} x {S}: classic {res["classic"][2]:.3f} ms  pipe {res["pipe"][2]:.3f} ms  status {res["pipe"][3]}  worst rel diff {worst:.2e}')
    print('   ' + ' '.join(lines), flush=True)
    dbg = ops.pipe_debug()
    if dbg is not None and os.environ.get('SUNERF_PIPE_DEBUG'):
        sel = dbg[0][:, 5] > 0
        d = dbg[0][sel]
        for l in sorted(set(d[:, 6].tolist()), reverse=True):
            m = d[:, 6] == l
            r = d[m].float()
            line = (f'   layer {l}: loop {r[:, 0].mean() / 100:.0f} us  in-spins {r[:, 1].mean():.0f} ({r[:, 2].mean() / 100:.0f} us)  '
                    f'out-spins {r[:, 3].mean():.0f} ({r[:, 4].mean() / 100:.0f} us)  chunks {r[:, 5].mean():.0f}')
            if l > 0 and dbg.shape[0] > 1:
                n = d[m][:, 5:6].float()
                q = dbg[1][sel][m].float() * 16 / n
                line += (f'\n        data wave 1 clocks/chunk: wait {q[:, 0].mean():.0f} barrier {q[:, 1].mean():.0f} k-steps {q[:, 2].mean():.0f} epilogue {q[:, 3].mean():.0f}'
                         f' (products + conversion {q[:, 4].mean():.0f}, + stores {q[:, 5].mean():.0f})')
                for w in (0, 1):
                    q = dbg[2 + w][sel][m].float() * 16 / n
                    line += f'\n        weight wave {4 + w} clocks/chunk: top {q[:, 0].mean():.0f} (wait {q[:, 3].mean():.0f} barrier {q[:, 4].mean():.0f}) reads+gate {q[:, 1].mean():.0f} matrix {q[:, 2].mean():.0f}'
            print(line)
        if dbg.shape[0] >= 6:      # timeline of iteration n_my / 2 in pipeline 0: shader clocks relative to the earliest stamp of the workgroup
            tl = dbg[4:6].reshape(-1).view(torch.int64).reshape(32, 8, 8)[:16]
            names_d = ['top', 'landed', 'barrier', 'k-steps', 'epilogue']
            names_w = ['top', 'landed', 'barrier', 'published', 'reads+DMA+gate', 'matrix']
            for b in range(16):
                t = tl[b]
                if int(t.max()) == 0:
                    continue
                t0 = int(t[t > 0].min())
                print(f'   workgroup {b} (layer {7 - b // 2 if b < 14 else 0}, half {b % 2}) at iteration n/2, clocks since its first stamp:')
                for w in range(8):
                    nm = names_d if w < 4 else names_w
                    print('        wave %d: ' % w + '  '.join(f'{nm[k]} {int(t[w, k]) - t0}' for k in range(len(nm)) if int(t[w, k]) > 0))


if __name__ == '__main__':
    cases = [(6, 40, 1), (6, 32, 1), (17, 128, 2), (64, 128, 3), (181, 128, 5)]
    if len(sys.argv) > 1:
        a = [int(v) for v in sys.argv[1:]]
        cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
    for n_side, S, reps in cases:
        run(n_side, S, reps)
