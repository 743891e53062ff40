"""AUTO policy calibration: probe value (FAST vs EXACT on the probe's rays, gate units) beside the true worst gate units of FAST
against the fp32 reference arithmetic, for hidden weights scaled x1 ... x4 and several probe sizes."""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, '2024-hl-spi3s-sunerf_amd')); sys.path.insert(0, os.path.join(R, 'oracle')); sys.path.insert(0, os.path.join(R, 'tests'))
import sunerf_oracle as orc
from sunerf_hip import ops
from test_gpu_precision import _scaled_case, oracle_pass, hip_pass, worst
for scale in (1.0, 2.0, 2.5, 3.0, 4.0):
    for seed in (3, 11):
        params, o, d, t, z = _scaled_case(scale, seed=seed)
        ref = oracle_pass(params, o, d, t, z)
        fast, _ = hip_pass(ops, params, o, d, t, z, ops.PRECISION_FAST)
        true_units = worst(ops, fast, ref, z.shape[1])
        row = []
        for rays in (64, 144):
            ops.PROBE_RAYS = rays
            pk = ops.PackedMLP([W.cuda() for W, _ in params], [b.cuda() for _, b in params], precision=ops.PRECISION_AUTO)
            ops.emission_render_fwd(pk, o.cuda(), d.cuda(), t.cuda(), z.cuda(), reg_radius=1.2)
            row.append(f'probe({rays}) {pk.last_probe:.3f}')
        print(f'hidden x {scale:g} seed {seed}: FAST true {true_units:.3f} gate units | ' + ' | '.join(row))
