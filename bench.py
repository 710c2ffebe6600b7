#!/usr/bin/env python3
"""Headline benchmark: sweep-steps/sec of the MPS two-site sweep on MNIST-shaped synthetic input.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One bench "step" = one pass of the hot path over one resident batch = what Network.train does per
batch (Network_class.py:327-333): forward (environment build) + one sweep of N-1 two-site steps.
Consecutive passes alternate sweep direction, as training does.  `value` = sweep steps per second
over the whole job: K * (N-1) * n_gpus_factor / wall time, inputs resident in HBM.

Workload (BASELINE.json configs[2]): N = 784 sites, D = 2, L = 2, bond 20,
batch 5000 per GPU (weak scaling: every rank sweeps its own 5000-sample shard and the bond gradient
is all-reduced over RCCL each step), softmax + full_cross_ent, T = 0.1, lr = 1e-3, wd = 1e-3 with
the L2 norm-environment regulariser on (the reference's default), fixed-bond truncation (the only
policy under which "bond 20" exists).  Synthetic pixels, 81 % zeros.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (N, M, b_per_gpu, L)
    'c2': (784, 10, 1000, 2),
    'c3': (784, 20, 5000, 2),
    'c4': (784, 20, 2500, 2),     # per-GPU share of batch 20000 over 8 GPUs (run with --gpus 8)
    'c5': (784, 50, 5000, 10),    # ten labels, bond 50: the large-tensor path of the step (kernels_big.hip)
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def synth(N, b, L, seed):
    rng = np.random.default_rng(seed)
    p = rng.random((b, N), dtype=np.float32) * (rng.random((b, N), dtype=np.float32) > 0.81)
    X = np.stack([np.sin(np.pi * p / 2), np.cos(np.pi * p / 2)], -1).astype(np.float32)
    y = rng.integers(0, L, b).astype(np.int32)
    return X, y


def init_cores(N, M, D, L, seed):
    from tensornetworkforml_amd.Network_class import random_canonical_cores
    rng = np.random.default_rng(seed)
    return random_canonical_cores(N, M, D, L, scale=float(M) * 0.5 * 0.64 * D, rng=rng)


def bytes_per_step(b, M, D, L):
    # read Lenv_{l-1}, write Lenv_l, read Renv_{l+2} (3 b M), x_{l-1}, x_l, x_{l+1}
    # (3 b D), f_prev + f_new (2 b L), y (b); float32
    return 4 * b * (3 * M + 3 * D + 2 * L + 1)


def cpu_baseline(N, M, D, L, b, n_steps, seed):
    """The float64 oracle (einsum/BLAS form with cached norm environments) timed on this host:
    one forward on the full batch + n_steps sweep steps; rate = (N-1) / (t_fwd + (N-1) t_step)."""
    from oracle import mps_oracle as mo
    X, y = synth(N, b, L, seed)
    X = X.astype(np.float64)
    rng = np.random.default_rng(seed + 1)
    cores = mo.random_cores(N, M, D, L, rng=rng, scale=float(M) * 0.5 * 0.64 * D)
    st = mo.MPSState(N, D, L, M, cores)
    mo.calibrate(st, X)
    t0 = time.perf_counter()
    f = mo.forward(st, X)
    t_fwd = time.perf_counter() - t0
    y1h = mo.one_hot(y, L)
    st.Lenv = {}
    # first step builds the norm-environment cache; time steady-state steps after it
    f = mo.sweep_step(st, f, y1h, 1e-3, 1e-3, True, False, 'softmax', 'full_cross_ent', 0.1, 'fixed')
    t0 = time.perf_counter()
    for _ in range(n_steps):
        f = mo.sweep_step(st, f, y1h, 1e-3, 1e-3, True, False, 'softmax', 'full_cross_ent', 0.1, 'fixed')
    t_step = (time.perf_counter() - t0) / n_steps
    rate = (N - 1) / (t_fwd + (N - 1) * t_step)
    return rate, t_fwd, t_step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='c3', choices=sorted(CONFIGS))
    ap.add_argument('--policy', default='fixed', choices=['fixed', 'reference'])
    ap.add_argument('--no-l2', action='store_true', help='weight decay as wd*B instead of the L2 norm term')
    ap.add_argument('--cpu-steps', type=int, default=40, help='oracle steps timed for cpu_baseline (0 = skip)')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--svd-stop', type=float, default=None, help='Jacobi stopping threshold (tnml_set_svd_stop); default: the library default')
    ap.add_argument('--no-cold', action='store_true', help='skip the re-initialised (cold start) passes')
    ap.add_argument('--sync-interval', type=int, default=0, help='drain the stream every so many sweep steps (runs under rocprofv3 --pmc)')
    ap.add_argument('--check-launches', action='store_true', help='read the launch status back after every kernel launch')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if rank == 0:
            print('warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
    N, M, b, L = CONFIGS[args.config]
    D = 2

    from tensornetworkforml_amd import _hip, dist as tdist
    dist = None
    if world > 1:
        dist = tdist.init_process_group(rank, world, 'gloo')     # rendezvous only; the data path is RCCL
    if _hip.device_count() <= local_rank:
        raise SystemExit('bench.py needs a gfx950 GPU per rank (visible: %d)' % _hip.device_count())
    ctx = _hip.Context(N, D, L, M, b, device=local_rank)
    if args.svd_stop is not None:
        ctx.set_svd_stop(args.svd_stop)
    tdist.attach_comm(ctx, rank, world)
    if args.sync_interval:
        ctx.set_sync_interval(args.sync_interval)
    if args.check_launches:
        ctx.debug_enable(4)

    X, y = synth(N, b, L, 1234 + rank)          # every rank owns a different shard
    cores = init_cores(N, M, D, L, 99)            # same cores on every rank
    ctx.set_input(X, y)

    def init_network():
        ctx.set_cores(cores, 0)
        # calibration on the same batch (Network_class.py:168-176); log-domain: max|f| ~ 1e-66 before it
        F2 = float(np.exp(ctx.forward_logabsmax() / N))
        ctx.scale_cores(1.0 / F2)

    init_network()

    hp = dict(lr=1e-3, weight_dec=1e-3, L2_flag=not args.no_l2, act_fn='softmax', loss_fn='full_cross_ent', T=0.1,
              trunc=args.policy)

    def one_pass(want=False):
        ctx.forward(want_f=False)
        left_dir = ctx.l_pos == N - 1
        return ctx.sweep(left_dir, N - 1, True, hp['lr'], hp['weight_dec'], hp['L2_flag'], hp['act_fn'],
                         hp['loss_fn'], hp['T'], hp['trunc'], want_metrics=want, want_f=want)

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.synchronize()

    for _ in range(args.warmup):
        one_pass()
    barrier()
    ctx.svd_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    sw_tot, n_svd, rounds_tot = ctx.svd_stats(reset=True)
    # SURVEY.md 8.4 break-down: the environment build (forward) and the host -> device hand-over of one batch
    # on their own, and the rate of the sweep alone
    ctx.synchronize()
    ctx.timer_start()
    ctx.forward(want_f=False)
    fwd_ms = ctx.timer_stop()
    left_dir = ctx.l_pos == N - 1
    ctx.sweep(left_dir, N - 1, True, hp['lr'], hp['weight_dec'], hp['L2_flag'], hp['act_fn'], hp['loss_fn'], hp['T'],
              hp['trunc'], want_metrics=False, want_f=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.set_input(X, y)
    ctx.synchronize()
    h2d_ms = 1e3 * (time.perf_counter() - t0)
    # validity: one more pass handing back metrics and f; non-finite values raise inside the library
    met, f = one_pass(want=True)
    finite = bool(np.isfinite(f).all() and np.isfinite(met).all())

    sweep_steps = args.steps * (N - 1)
    # weak scaling: every rank performs the same sweep steps on its own shard of b samples, so the job's work unit is
    # "one sweep step over one b-sample shard" and the whole-job aggregate is world x the common step rate
    steps_per_s = sweep_steps * 1.0 / dt
    value = world * steps_per_s
    out = {
        'metric': 'sweep-steps/sec, 28x28 MNIST-shaped, bond=%d, batch=%d per GPU' % (M, b),
        'value': value,
        'unit': 'sweep-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * dt / args.steps,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': '%s: N=784 sites, D=2, L=%d, bond %d, batch %d/GPU, softmax+full_cross_ent, '
                               'trunc=%s, L2_flag=%s; pass = forward + %d-step sweep'
                               % (args.config, L, M, b, args.policy, hp['L2_flag'], N - 1),
                   'global_batch': b * world, 'sweep_steps_per_pass': N - 1, 'parallelism': 'dp%d' % world},
        'value_definition': 'sweep steps per second x number of %d-sample shards (= GPUs) stepping in lock-step; '
                            'global-batch sweep steps per second: %.1f' % (b, steps_per_s),
        'finite': finite,
        'breakdown': {'forward_ms': fwd_ms, 'h2d_batch_ms': h2d_ms,
                      'sweep_only_steps_per_s': (N - 1) / max(1e-3 * (1e3 * dt / args.steps - fwd_ms), 1e-9),
                      'steps_per_s_incl_h2d': (N - 1) / (dt / args.steps + 1e-3 * h2d_ms)},
        'final_accuracy': float(met[-1, 0]),
        # the SVD is iterative: how much work the timed passes actually contained
        'jacobi': {'sweeps_per_svd': sw_tot / max(n_svd, 1), 'rounds_per_svd': rounds_tot / max(n_svd, 1),
                   'svd_stop2': args.svd_stop if args.svd_stop is not None else 1e-6},
    }

    if rank == 0 and not args.no_kernel_profile:
        # per-kernel device time with HIP events on the library's own stream (two more passes)
        ctx.profile_reset()
        ctx.profile_enable(True)
        one_pass()
        one_pass()
        ctx.profile_enable(False)
        # the wide kernel is the one-shot MFMA formulation whenever its tile operands fit LDS (all configs but c5, which
        # streams the merged tensor through LDS in chunks); on one GPU
        # the slab reduction rides inside the narrow launch (helper workgroups), so reduce_slabs_kernel shows 0 launches
        wide_name = 'wide_step_mfma_tiled_kernel' if args.config == 'c5' else 'wide_step_mfma_kernel'
        names = ['env_chain_kernel', wide_name, 'reduce_slabs_kernel', 'narrow_step_kernel']
        kern = {}
        for i, nm in enumerate(names):
            ms, n = ctx.profile_get(i)
            kern[nm] = {'avg_us': 1e3 * ms / max(n, 1), 'launches': n, 'total_ms': ms}
        out['kernels'] = kern
        bstep = bytes_per_step(b, M, D, L)
        wide_us = kern[wide_name]['avg_us']
        ach = bstep / (wide_us * 1e-6) / 1e9
        # HBM traffic of that kernel from committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE
        # runs of this command; gfx950 correction: FETCH_SIZE counts half of a coalesced read stream)
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'r01_pmc_%s.json' % args.config)
        if os.path.exists(pmc) and args.policy == 'fixed' and not args.no_l2:
            w = json.load(open(pmc)).get(wide_name, {})
            if 'FETCH_SIZE' in w and 'WRITE_SIZE' in w:
                traffic = (2.0 * w['FETCH_SIZE']['mean_KB'] + w['WRITE_SIZE']['mean_KB']) * 1024.0
        # the same kernel's average in the committed rocprofv3 --kernel-trace --stats summary of this command (the HIP
        # event figure above brackets one isolated launch and carries its ~3 us of launch latency)
        rocprof_us = None
        kst = os.path.join(ROOT, 'profiles', 'r01_kernel_stats_%s.csv' % args.config)
        if os.path.exists(kst) and args.policy == 'fixed' and not args.no_l2:     # the summary is of the default case
            import csv
            for row in csv.DictReader(open(kst)):
                if wide_name + '(' in row['Name'] or row['Name'].split('(')[0].endswith(wide_name):
                    rocprof_us = float(row['AverageNs']) / 1e3
        flops_step = 4.0 * b * D * D * M * M * L + 2.0 * b * D * M * M      # dB GEMM + f GEMM + env extension
        if args.config == 'c5':
            # bond 50 / ten labels is the MFMA-bound case (580 flop per algorithmic byte); float32 MFMA peak of
            # MI355X_MICROARCH.md: 157.3 TFLOP/s (v_mfma_f32_16x16x4_f32 runs at the float32 vector rate)
            tf = flops_step / (wide_us * 1e-6) / 1e12
            out['roofline'] = {'bound': 'mfma', 'kernel': wide_name, 'achieved': tf, 'peak': 157.3, 'unit': 'TFLOP/s',
                               'frac': tf / 157.3, 'traffic': None, 'algorithmic_flops_per_launch': flops_step,
                               'algorithmic_bytes_per_launch': bstep, 'kernel_avg_us_hip_events': wide_us,
                               'kernel_avg_us_rocprofv3': None}
        else:
            out['roofline'] = {'bound': 'hbm', 'kernel': wide_name, 'achieved': ach, 'peak': HBM_PEAK_GBS,
                               'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, 'traffic': traffic,
                               'algorithmic_bytes_per_launch': bstep, 'kernel_avg_us_hip_events': wide_us,
                               'kernel_avg_us_rocprofv3': rocprof_us,
                               'whole_step_GBs': bstep * steps_per_s / 1e9}
    elif dist is not None and not args.no_kernel_profile:
        # keep the collectives of the profiling passes matched on every rank
        one_pass()
        one_pass()

    if not args.no_cold:
        # "cold" passes: network re-initialised (random cores, calibrated) and swept twice -- the regime
        # of the first training batches, where the merged tensors are far from their SVD form and the
        # Jacobi iteration needs its full 7-9 sweeps
        init_network()
        barrier()
        ctx.svd_stats(reset=True)
        t0 = time.perf_counter()
        one_pass()
        one_pass()
        barrier()
        dtc = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([dtc], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtc = float(t.item())
        sw_c, n_c, _ = ctx.svd_stats(reset=True)
        out['cold_start'] = {'value': 2 * (N - 1) / dtc, 'unit': 'sweep-steps/s', 'passes': 2,
                             'jacobi_sweeps_per_svd': sw_c / max(n_c, 1)}

    if rank == 0 and args.cpu_steps > 0:
        rate, t_fwd, t_step = cpu_baseline(N, M, D, L, b, args.cpu_steps, 1234)
        try:
            ncpu = len(os.sched_getaffinity(0))
        except Exception:
            ncpu = os.cpu_count()
        blas_threads = None
        try:                                   # threads NumPy's BLAS actually runs the einsum / matmul calls on
            from threadpoolctl import threadpool_info
            blas_threads = max([int(i.get('num_threads', 1)) for i in threadpool_info()] or [1])
        except Exception:
            pass
        out['cpu_baseline'] = {'value': rate, 'unit': 'sweep-steps/s', 'cores': blas_threads or ncpu, 'host_cpus': ncpu,
                               'kind': 'port',
                               'sample': 'float64 NumPy oracle (einsum/BLAS, cached norm environments): 1 forward on '
                                         'the full %d-sample batch (%.2f s) + %d sweep steps (%.1f ms each), '
                                         'extrapolated to a %d-step pass' % (b, t_fwd, args.cpu_steps, 1e3 * t_step, N - 1)}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
