#!/usr/bin/env python3
"""Builds profiles/hbm_traffic.json from rocprofv3 PMC passes (development aid).

usage: hbm_traffic.py <out.json> <mode> <rays> <samples> <d_filter> <fetch_dir> <write_dir>
  fetch_dir / write_dir: output directories of `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs (separate
  passes, --output-format csv) of the same `bench.py --mode <mode>` command.
Per kernel the LAST dispatch is taken (steady state); FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950
(wide coalesced reads are tallied at half their size); both counters are in KiB."""
import csv
import glob
import json
import os
import sys

KERNELS = ('render_fwd_kernel', 'integral_bwd', 'dgrad', 'wgrad_kernel', 'bwd_pipe_kernel', 'bwd_prologue_kernel', 'reduce_grads',
           'loss_kernel', 'adam_kernel', 'grad_norm_kernel', 'pack_mlp')


def last_per_kernel(directory, counter):
    rows = {}
    paths = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    for path in sorted(paths, key=os.path.getmtime)[-1:]:      # the newest run only (older runs may share the directory)
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get('Counter_Name') != counter:
                    continue
                name = r['Kernel_Name']
                key = next((k for k in KERNELS if k in name), None)
                if key is None:
                    continue
                did = int(r['Dispatch_Id'])
                if key not in rows or did >= rows[key][0]:
                    rows[key] = (did, float(r['Counter_Value']))
    return {k: v[1] for k, v in rows.items()}


def main():
    out, mode, rays, samples, d_filter, fdir, wdir = sys.argv[1:8]
    fetch, write = last_per_kernel(fdir, 'FETCH_SIZE'), last_per_kernel(wdir, 'WRITE_SIZE')
    per_kernel = {k: (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0 for k in sorted(set(fetch) | set(write))}
    data = {}
    if os.path.exists(out):
        with open(out) as f:
            data = json.load(f)
    data['note'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), last dispatch of each kernel; FETCH_SIZE '
                    'doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); bytes')
    data[f'{mode}_bytes_per_step'] = sum(per_kernel.values())
    data[f'{mode}_bytes_per_kernel'] = per_kernel
    data[f'{mode}_kernel_bytes'] = per_kernel.get('render_fwd_kernel')      # the fused render pass (bench.py: roofline.traffic)
    data[f'{mode}_bwd_kernel_bytes'] = per_kernel.get('bwd_pipe_kernel')     # the pipelined backward kernel, when it ran
    data[f'{mode}_fetch_kib_raw'] = fetch
    data[f'{mode}_write_kib_raw'] = write
    data[f'{mode}_config'] = {'rays': int(rays), 'samples': int(samples), 'd_filter': int(d_filter)}
    for k in (f'{mode}_workload',):
        data.pop(k, None)
    with open(out, 'w') as f:
        json.dump(data, f, indent=1)
    print(json.dumps({k: round(v / 1e9, 3) for k, v in per_kernel.items()}), 'total GB', round(sum(per_kernel.values()) / 1e9, 2))


if __name__ == '__main__':
    main()
