#!/usr/bin/env python3
"""Headline benchmark: sweep-steps/sec of the MPS two-site sweep on MNIST-shaped synthetic input.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One bench "step" = one training pass of the hot path = what Network.train does per batch (Network_class.py:324-333):
a NEW batch becomes resident, forward (environment build), one sweep of N-1 two-site steps; consecutive passes
alternate sweep direction.  The passes cycle through `--batches` (default 4) distinct synthetic batches that were staged
in HBM before the timed region (tnml_stage_batch); inside the timed region a pass starts with the device-side hand-over
of its batch (tnml_select_batch: re-tiling + label copy, no host traffic).  `value` = sweep steps per second over the
whole job, inputs resident in HBM, barrier + device synchronisation on both sides, max over ranks.

Workload (BASELINE.json configs[2]): N = 784 sites, D = 2, L = 2, bond 20, batch 5000 per GPU,
softmax + full_cross_ent, T = 0.1, lr = 1e-3, wd = 1e-3 with the L2 norm-environment regulariser on (the reference's
default), fixed-bond truncation (the only policy under which "bond 20" exists).  Synthetic pixels,
81 % zeros.  --config c4 is the fixed global batch of 20000 split over the ranks (strong scaling).

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
    # name: (N, M, batch, L, batch_is_global)
    'c2': (784, 10, 1000, 2, False),
    'c3': (784, 20, 5000, 2, False),
    'c4': (784, 20, 20000, 2, True),    # BASELINE.json configs[3]: batch 20000 sharded over the ranks (8 in the baseline)
    'c5': (784, 50, 5000, 10, False),   # ten labels, bond 50: the large-tensor path of the step (kernels_big.hip)
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3


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


def flops_per_step(b, M, D, L):
    # dB GEMM + f GEMM + environment extension
    return 4.0 * b * D * D * M * M * L + 2.0 * b * D * M * M


def host_threads():
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count()
    blas = None
    try:                                   # threads NumPy's BLAS actually runs the einsum / matmul calls on
        from threadpoolctl import threadpool_info
        blas = max([int(i.get('num_threads', 1)) for i in threadpool_info()] or [1])
    except Exception:
        pass
    return ncpu, blas


def cpu_baseline(N, M, D, L, b, n_steps, seed):
    """B4 of BASELINE.md: the float64 oracle (einsum/BLAS form with cached norm environments) timed on this host:
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
    hp = (1e-3, 1e-3, True, False, 'softmax', 'full_cross_ent', 0.1, 'fixed')
    for _ in range(8):          # past the ramp of the behind bond (1, 2, 4, 8, 16, 20): steady mid-chain steps
        f = mo.sweep_step(st, f, y1h, *hp)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        f = mo.sweep_step(st, f, y1h, *hp)
    t_step = (time.perf_counter() - t0) / n_steps
    rate = (N - 1) / (t_fwd + (N - 1) * t_step)
    return rate, t_fwd, t_step, st, f, X, y1h


def cpu_baseline_reference_form(N, M, D, L, st, f, y1h, budget_s):
    """B1 / B2 / B3 of BASELINE.md section 3: the step in the reference's own computational form
    (oracle/mps_reference_form.py: broadcast-multiply-sum contractions, L2 norm environments rebuilt from the chain ends
    at every step, float64).  B3 and the `first_sweep` forms continue from the mid-chain fixed-bond state the B4 timing left
    behind; the `steady_state` forms step a chain whose bonds are all 2 (what the reference's rule leaves after one sweep:
    BASELINE.md section 2 quotes 4.2 steps/s for it).  A few steps each, bounded by
    `budget_s` seconds in total; rates are sweep steps per second of the step alone (the reference's forward, 84-112 s at
    this shape in the survey container, is not included)."""
    import copy
    from oracle import mps_oracle as mo
    from oracle import mps_reference_form as rf
    out = {}
    t_end = time.perf_counter() + budget_s
    for tag, trunc, l2 in (('B3_fixed_bond_L2', 'fixed', True), ('B1_reference_truncation_L2_first_sweep', 'reference', True),
                           ('B2_reference_truncation_noL2_first_sweep', 'reference', False)):
        s2 = copy.deepcopy(st)
        s2.Ln, s2.Rn = {}, {}
        ff = f.copy()
        times = []
        for i in range(4):
            if time.perf_counter() > t_end and times:
                break
            t0 = time.perf_counter()
            ff = rf.sweep_step(s2, ff, y1h, 1e-3, 1e-3, l2, False, 'softmax', 'full_cross_ent', 0.1, trunc)
            times.append(time.perf_counter() - t0)
        # the reference-truncation variants continue from the fixed-bond state: the rule keeps m = left bond, so the bond
        # behind stays at its incoming value (20 at c3) -- the regime of the reference's FIRST sweep over a bond-M network
        use = times[1:] if len(times) > 1 else times
        out[tag] = {'steps_per_s': 1.0 / float(np.mean(use)), 'ms_per_step': 1e3 * float(np.mean(use)), 'steps_timed': len(use),
                    'behind_bond': int(s2.ml(s2.l_pos)), 'ahead_bond': int(s2.mr(min(s2.l_pos + 1, N - 1)))}
    # the reference's STEADY state: after its first sweep every bond is 2 (m = left bond, 2 at the chain start); a chain of
    # bond-2 cores, environments from the oracle's forward, the first steps of a right sweep
    Xb = st.X
    rng = np.random.default_rng(7)
    for tag, l2 in (('B1_reference_truncation_L2_steady_state', True), ('B2_reference_truncation_noL2_steady_state', False)):
        if time.perf_counter() > t_end + 10.0:
            break
        s3 = mo.MPSState(N, D, L, 2, mo.random_cores(N, 2, D, L, rng=rng, scale=2 * 0.5 * 0.64 * D))
        mo.calibrate(s3, Xb[:256])
        ff = mo.forward(s3, Xb)
        times = []
        for i in range(4):
            t0 = time.perf_counter()
            ff = rf.sweep_step(s3, ff, y1h, 1e-3, 1e-3, l2, False, 'softmax', 'full_cross_ent', 0.1, 'reference')
            times.append(time.perf_counter() - t0)
        use = times[1:]
        out[tag] = {'steps_per_s': 1.0 / float(np.mean(use)), 'ms_per_step': 1e3 * float(np.mean(use)), 'steps_timed': len(use),
                    'behind_bond': int(s3.ml(s3.l_pos)), 'ahead_bond': int(s3.mr(min(s3.l_pos + 1, N - 1)))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)      # the driver's command: --steps 20 --warmup 5
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='c3', choices=sorted(CONFIGS))
    ap.add_argument('--policy', default='fixed', choices=['fixed', 'reference'])
    ap.add_argument('--no-l2', action='store_true', help='weight decay as wd*B instead of the L2 norm term')
    ap.add_argument('--batches', type=int, default=4, help='distinct staged batches the passes cycle through (1: one resident batch)')
    ap.add_argument('--cpu-steps', type=int, default=40, help='oracle steps timed for cpu_baseline (0 = skip)')
    ap.add_argument('--ref-form-budget', type=float, default=25.0, help='seconds for cpu_baseline_reference_form (0 = skip)')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--svd-stop', type=float, default=None, help='Jacobi stopping threshold (tnml_set_svd_stop); default: the library default')
    ap.add_argument('--no-cold', action='store_true', help='skip the re-initialised (cold start) passes')
    ap.add_argument('--no-resident', action='store_true', help='skip the secondary single-resident-batch measurement')
    ap.add_argument('--classic', action='store_true', help='classic launch sequence instead of the pipelined single-launch step')
    ap.add_argument('--per-step', action='store_true', help='one launch per sweep step (round 2) instead of the persistent per-sweep launch')
    ap.add_argument('--persist-mode', type=int, default=None, choices=[1, 2], help='persistent sweep as one kernel (1) or one kernel per role (2); default: the library default')
    ap.add_argument('--pipe-tiles', type=int, default=0, help='sample tiles per batch-side workgroup on steps with a long SVD (tnml_set_step_pipeline(ctx, n), n >= 2)')
    ap.add_argument('--sync-interval', type=int, default=0, help='drain the stream every so many sweep steps (runs under rocprofv3 --pmc)')
    ap.add_argument('--check-launches', action='store_true', help='read the launch status back after every kernel launch')
    ap.add_argument('--event-handoffs', action='store_true', help='hand-offs between the two streams of a context as events instead of sequence numbers in memory (runs under rocprofv3 --pmc, which serialises dispatches: a polling kernel would wait for a producer that cannot start)')
    ap.add_argument('--no-comm-overlap', action='store_true', help='multi-GPU: one fused launch per step with the all-reduce between launches (round 2) instead of update side / batch side + all-reduce on two streams')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
    N, M, b_cfg, L, is_global = CONFIGS[args.config]
    b = b_cfg // world if is_global else b_cfg          # per-rank shard
    D = 2

    from tensornetworkforml_amd import _hip, dist as tdist
    dist = None
    if world > 1:
        dist = tdist.init_process_group(rank, world)     # rendezvous only; the data path is RCCL
    if _hip.device_count() <= local_rank:
        raise SystemExit('bench.py needs a gfx950 GPU per rank (visible: %d)' % _hip.device_count())
    ctx = _hip.Context(N, D, L, M, b, device=local_rank)
    if args.svd_stop is not None:
        ctx.set_svd_stop(args.svd_stop)
    # RCCL prints a version banner on file descriptor 1 when a communicator is created; this program's stdout is ONE JSON line
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        tdist.attach_comm(ctx, rank, world)
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    if args.no_comm_overlap:
        ctx.set_comm_overlap(False)
    if args.event_handoffs:
        ctx.set_flag_handoffs(False)
    if args.sync_interval:
        ctx.set_sync_interval(args.sync_interval)
    if args.check_launches:
        ctx.debug_enable(4)
    if args.per_step:
        ctx.set_persistent(0)
    elif args.persist_mode:
        ctx.set_persistent(args.persist_mode)
    if args.classic:
        ctx.set_step_pipeline(False)
    elif args.pipe_tiles >= 2:
        ctx.set_step_pipeline(args.pipe_tiles)

    nb = max(1, min(args.batches, 8))
    batches = [synth(N, b, L, 1234 + 97 * k + rank) for k in range(nb)]     # every rank owns different shards
    for k, (Xk, yk) in enumerate(batches):
        ctx.stage_batch(k, Xk, yk)
    cores = init_cores(N, M, D, L, 99)            # same cores on every rank
    ctx.select_batch(0)

    def init_network():
        ctx.set_cores(cores, 0)
        # calibration on the first batch (Network_class.py:168-176); log-domain: max|f| ~ 1e-66 before it
        F2 = float(np.exp(ctx.forward_logabsmax() / N))
        ctx.scale_cores(1.0 / F2)

    init_network()
    hp = dict(lr=1e-3, weight_dec=1e-3, L2_flag=not args.no_l2, act_fn='softmax', loss_fn='full_cross_ent', T=0.1,
              trunc=args.policy)
    counter = [0]
    # device time of EVERY sweep of this process (for comparison with a rocprofv3 --kernel-trace --stats summary of the same
    # command, whose per-kernel average runs over all launches, not only the timed passes)
    whole = {'ms': 0.0, 'launches': 0}

    def drain_whole():
        ms, _ = ctx.profile_get(4)
        _, nl = ctx.profile_get(5)
        whole['ms'] += ms
        whole['launches'] += nl
        ctx.profile_reset()

    def one_pass(want=False, rotate=True):
        if rotate and nb > 1:
            ctx.select_batch(counter[0] % nb)
            counter[0] += 1
        ctx.forward(want_f=False)
        left_dir = ctx.l_pos == N - 1
        return ctx.sweep(left_dir, N - 1, True, hp['lr'], hp['weight_dec'], hp['L2_flag'], hp['act_fn'],
                         hp['loss_fn'], hp['T'], hp['trunc'], want_metrics=want, want_f=want)

    def barrier():
        ctx.synchronize()          # also reads the device status word: a failed SVD inside the passes raises here
        if dist is not None:
            dist.barrier()
        ctx.synchronize()

    def timed(n_pass, **kw):
        barrier()
        ctx.svd_stats(reset=True)
        drain_whole()
        ctx.profile_enable(2)      # one event pair per sweep call on the library's stream; nothing waits inside
        t0 = time.perf_counter()
        for _ in range(n_pass):
            one_pass(**kw)
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            dt = dist.max_float(dt)
        sweep_ms, n_launch = ctx.profile_get(4)
        cnt = ctx.counters()
        _, n_steps_pipe = ctx.profile_get(5)
        drain_whole()
        sw_tot, n_svd, rounds_tot = ctx.svd_stats(reset=True)
        chol = ctx.cholesky_steps
        return dict(dt=dt, sweep_ms=sweep_ms, launches=n_launch, pipe_steps=n_steps_pipe, counters=cnt,
                    sweeps_per_svd=sw_tot / max(n_svd, 1), rounds_per_svd=rounds_tot / max(n_svd, 1),
                    cholesky_fraction=chol / max(n_svd, 1))

    # phase markers (tnml_marker: an empty kernel of `id` workgroups) cut a rocprofv3 trace of this command to one phase
    # (tools/rocprof_summary.py): 1 warm-up | 2 timed passes | 3 break-down + per-kernel passes | 4 resident | 5 | 6 cold | 7
    ctx.profile_reset()
    ctx.profile_enable(2)
    ctx.marker(1)
    for _ in range(args.warmup):
        one_pass()
    ctx.marker(2)
    main_run = timed(args.steps)
    ctx.marker(3)
    dt = main_run['dt']

    # SURVEY.md 8.4 break-down: the environment build (forward) and the host -> device hand-over of one batch on their own
    ctx.synchronize()
    ctx.timer_start()
    ctx.select_batch(0)
    sel_ms = ctx.timer_stop()
    ctx.timer_start()
    ctx.forward(want_f=False)
    fwd_ms = ctx.timer_stop()
    left_dir = ctx.l_pos == N - 1
    ctx.sweep(left_dir, N - 1, True, hp['lr'], hp['weight_dec'], hp['L2_flag'], hp['act_fn'], hp['loss_fn'], hp['T'],
              hp['trunc'], want_metrics=False, want_f=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.set_input(*batches[0])
    ctx.synchronize()
    h2d_ms = 1e3 * (time.perf_counter() - t0)
    # the exchange of a multi-GPU step on its own: one all-reduce of the pre-gradient message (collective: every rank calls it)
    z_floats = min((M * D) * D * D * M * L, 2 * M * D * D * M * L) + 4
    allreduce_us = ctx.comm_probe(z_floats) if (world > 1 or os.environ.get('TNML_FORCE_COMM') == '1') else None
    # validity: one more pass handing back metrics and f; non-finite values raise inside the library
    met, f = one_pass(want=True)
    finite = bool(np.isfinite(f).all() and np.isfinite(met).all())

    sweep_steps = args.steps * (N - 1)
    steps_per_s = sweep_steps / dt
    strong = is_global
    # weak scaling (default): every rank performs the same sweep steps on its own b-sample shard, the job's work unit is
    # "one sweep step over one shard" and the whole-job aggregate is world x the common step rate.  c4 (strong): the global
    # batch is fixed, the job performs `steps_per_s` sweep steps per second on it whatever the rank count.
    value = steps_per_s if strong else world * steps_per_s
    bstep = bytes_per_step(b, M, D, L)
    out = {
        'metric': 'sweep-steps/sec, 28x28 MNIST-shaped, bond=%d, batch=%d%s' % (M, b_cfg, ' (global)' if strong else ' per GPU'),
        'value': value,
        'unit': 'sweep-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * dt / args.steps,
        'higher_is_better': True,
        'scaling': 'strong' if strong else 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': '%s: N=784 sites, D=2, L=%d, bond %d, batch %d/GPU, softmax+full_cross_ent, trunc=%s, L2_flag=%s; '
                               'pass = hand-over of one of %d staged batches + forward + %d-step sweep%s'
                               % (args.config, L, M, b, args.policy, hp['L2_flag'], nb, N - 1, ' (classic launch sequence)' if args.classic else ''),
                   'global_batch': b * world, 'sweep_steps_per_pass': N - 1, 'parallelism': 'dp%d' % world,
                   'distinct_batches': nb},
        'value_definition': ('global-batch (%d) sweep steps per second' % (b * world)) if strong else
                            ('sweep steps per second x number of %d-sample shards (= GPUs) stepping in lock-step; common step '
                             'rate (= sweep steps per second on the global batch of %d): %.1f' % (b, b * world, steps_per_s)),
        'global_batch_steps_per_s': steps_per_s,
        'weak_scaling_aggregate': world * steps_per_s,
        'finite': finite,
        'breakdown': {'forward_ms': fwd_ms, 'select_batch_ms': sel_ms, 'h2d_batch_ms': h2d_ms,
                      'sweep_only_steps_per_s': sweep_steps / max(1e-3 * main_run['sweep_ms'], 1e-9),
                      'steps_per_s_incl_h2d': (N - 1) / (dt / args.steps + 1e-3 * h2d_ms),
                      # multi-GPU: device time of ONE all-reduce of the step's message (%d floats) issued back to back on the
                      # exchange stream; inside a sweep it runs beside the SVD of the step (tnml_set_comm_overlap)
                      'allreduce_us': allreduce_us, 'allreduce_floats': z_floats if allreduce_us is not None else None,
                      'comm_overlap': (not args.no_comm_overlap) if allreduce_us is not None else None},
        'final_accuracy': float(met[-1, 0]),
        # the library's own account of the timed passes (tnml_get_counters: from the bond dimensions of every step that ran)
        'counters': main_run['counters'],
        # the SVD is iterative: how much work the timed passes actually contained
        'jacobi': {'sweeps_per_svd': main_run['sweeps_per_svd'], 'rounds_per_svd': main_run['rounds_per_svd'],
                   'cholesky_fraction': main_run['cholesky_fraction'],
                   'svd_stop2': args.svd_stop if args.svd_stop is not None else 1e-6},
    }

    # ---- roofline of the dominant kernel, measured live over the timed region ------------------------------------------------
    # Pipelined step: one launch of step_pipe_kernel per sweep step carries the whole step (update + SVD of step k next to f /
    # pre-gradient of step k+1), so it is the dominant kernel and its average duration is the device time of the timed
    # sweeps (HIP events on the library's stream, first to last launch of each sweep) / launches.  Classic sequence and the
    # large-tensor path (c5): several kernels per step; the figure is then the whole step per launch group.
    # (c5: only the steps next to the chain ends fit the single-launch kernel; the figure there is per sweep step)
    all_pipe = main_run['pipe_steps'] >= sweep_steps
    step_launches = main_run['pipe_steps'] if all_pipe else sweep_steps
    step_us = 1e3 * main_run['sweep_ms'] / max(step_launches, 1)
    if args.config == 'c5':
        fl = flops_per_step(b, M, D, L)
        tf = fl / (step_us * 1e-6) / 1e12
        out['roofline'] = {'bound': 'mfma', 'kernel': 'sweep step (wide_step_mfma_tiled_kernel + large-tensor chain; pipelined launches at the chain ends)',
                           'achieved': tf, 'peak': MFMA_F32_PEAK_TF, 'unit': 'TFLOP/s', 'frac': tf / MFMA_F32_PEAK_TF, 'traffic': None,
                           'algorithmic_flops_per_step': fl, 'algorithmic_bytes_per_step': bstep, 'step_avg_us_hip_events': step_us}
    else:
        ach = bstep / (step_us * 1e-6) / 1e9
        persistent = all_pipe and main_run['launches'] <= 2 * args.steps
        out['roofline'] = {'bound': 'hbm', 'kernel': ('sweep_persist_kernel (ONE launch per sweep of %d steps; the figures are per step)' % (N - 1)) if persistent
                           else ('step_pipe_kernel' if all_pipe else 'wide_step_mfma_kernel + narrow_step_kernel'),
                           'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, 'traffic': None,
                           'algorithmic_bytes_per_launch': bstep * ((N - 1) if persistent else 1), 'algorithmic_bytes_per_step': bstep,
                           'kernel_avg_us_hip_events': step_us * ((N - 1) if persistent else 1), 'step_avg_us_hip_events': step_us,
                           'launches_timed': main_run['launches'] if persistent else step_launches, 'steps_timed': step_launches,
                           'note': 'one launch = one whole sweep (persistent kernel) or one whole sweep step; the step is bound by the sequential SVD of its merged tensor '
                                   '(critical_path below), not by HBM'}
        # HBM traffic from the committed counter passes (separate rocprofv3 --pmc runs of this command, cut to the timed passes by
        # the phase markers; never collected inside a bench run).  gfx950: FETCH_SIZE counts half of a coalesced read stream
        # (MI355X_MICROARCH.md, HBM section): x 2.
        pmc = os.path.join(ROOT, 'profiles', 'r03_pmc_%s.json' % args.config)
        if os.path.exists(pmc) and args.policy == 'fixed' and not args.no_l2 and not args.classic:
            allk = json.load(open(pmc))
            scope = 'timed' if 'timed' in allk else 'whole_process'
            kname = 'sweep_persist_kernel' if persistent else 'step_pipe_kernel'
            w = allk.get(scope, {}).get(kname, {})
            if 'FETCH_SIZE' in w and 'WRITE_SIZE' in w:
                per_launch = (2.0 * w['FETCH_SIZE']['mean_KB'] + w['WRITE_SIZE']['mean_KB']) * 1024.0
                out['roofline']['traffic_from_profiles'] = {
                    'bytes_per_launch': per_launch, 'bytes_per_step': per_launch / ((N - 1) if persistent else 1),
                    'source': 'profiles/r03_pmc_%s.json [%s][%s] (separate rocprofv3 --pmc passes, not this run)' % (args.config, scope, kname)}

    if rank == 0 and not args.no_kernel_profile:
        # per-kernel device time with HIP events around every launch (two more passes; synchronises after each launch)
        drain_whole()
        ctx.profile_enable(1)
        one_pass()
        one_pass()
        ctx.profile_enable(2)
        names = ['env_chain_kernel', 'batch-side kernel (classic wide kernel / prologue of a pipelined sweep)', 'reduce_slabs_kernel',
                 'step kernel (step_pipe_kernel, or the classic narrow kernel / large-tensor chain)']
        kern = {}
        for i, nm in enumerate(names):
            ms, n = ctx.profile_get(i)
            kern[nm] = {'avg_us': 1e3 * ms / max(n, 1), 'launches': n, 'total_ms': ms}
        out['kernels'] = kern
        # critical path of the update + SVD workgroup of a mid-chain step, from its in-kernel cycle stamps
        try:
            ctx.debug_enable(2)
            ctx.forward(want_f=False)
            left_dir = ctx.l_pos == N - 1
            ctx.sweep(left_dir, (N - 1) // 2, True, hp['lr'], hp['weight_dec'], hp['L2_flag'], hp['act_fn'], hp['loss_fn'], hp['T'],
                      hp['trunc'], want_metrics=False, want_f=False)
            ctx.synchronize()
            sc = ctx.step_debug('scalars')
            ctx.sweep(left_dir, N - 1 - (N - 1) // 2, False, hp['lr'], hp['weight_dec'], hp['L2_flag'], hp['act_fn'], hp['loss_fn'],
                      hp['T'], hp['trunc'], want_metrics=False, want_f=False)
            ctx.debug_enable(0 if not args.check_launches else 4)
            if sc[10] > 1:
                rounds = sc[55] if len(sc) > 55 and sc[55] > 0 else sc[9] * (sc[10] - 1)      # rounds actually run
                out['critical_path'] = {'what': 'workgroup 0 of one mid-chain step (shader cycles from s_memtime stamps)',
                                        'cycles_before_svd': sc[5], 'cycles_jacobi': sc[6], 'cycles_after_svd': sc[7],
                                        'jacobi_sweeps': sc[9], 'matrix_side': sc[10], 'rounds': rounds,
                                        'cycles_per_round': sc[6] / max(rounds, 1),
                                        'kernel_us_realtime_counter': sc[8] / 100.0,
                                        'clock_GHz': (sc[5] + sc[6] + sc[7]) / max(sc[8] * 10.0, 1e-9)}
        except Exception as e:      # the stamps are diagnostics: never fail the bench on them
            out['critical_path'] = {'error': str(e)[:200]}
    elif dist is not None and not args.no_kernel_profile:
        # keep the collectives of the profiling passes matched on every rank
        one_pass()
        one_pass()
        ctx.forward(want_f=False)
        left_dir = ctx.l_pos == N - 1
        ctx.sweep(left_dir, (N - 1) // 2, True, hp['lr'], hp['weight_dec'], hp['L2_flag'], hp['act_fn'], hp['loss_fn'], hp['T'], hp['trunc'],
                  want_metrics=False, want_f=False)
        ctx.sweep(left_dir, N - 1 - (N - 1) // 2, False, hp['lr'], hp['weight_dec'], hp['L2_flag'], hp['act_fn'], hp['loss_fn'], hp['T'],
                  hp['trunc'], want_metrics=False, want_f=False)

    if not args.no_resident and nb > 1:
        # the round-1 regime for comparison: the same passes on ONE resident batch (the chain converges onto it and the
        # Jacobi iteration settles at ~2 sweeps per SVD)
        ctx.select_batch(0)
        for _ in range(max(args.warmup, 2)):
            one_pass(rotate=False)
        ctx.marker(4)
        r = timed(args.steps, rotate=False)
        ctx.marker(5)
        out['resident_batch'] = {'value': (1 if strong else world) * sweep_steps / r['dt'], 'unit': 'sweep-steps/s',
                                 'jacobi_sweeps_per_svd': r['sweeps_per_svd'], 'cholesky_fraction': r['cholesky_fraction']}

    if not args.no_cold:
        # "cold" passes: network re-initialised (random cores, calibrated) and swept twice -- the regime of the first
        # training batches, where the merged tensors are far from their SVD form
        ctx.select_batch(0)
        init_network()
        ctx.marker(6)
        r = timed(2)
        ctx.marker(7)
        out['cold_start'] = {'value': (1 if strong else world) * 2 * (N - 1) / r['dt'], 'unit': 'sweep-steps/s', 'passes': 2,
                             'jacobi_sweeps_per_svd': r['sweeps_per_svd'], 'cholesky_fraction': r['cholesky_fraction']}

    if rank == 0 and args.cpu_steps > 0:
        rate, t_fwd, t_step, st, fcpu, Xc, y1h = cpu_baseline(N, M, D, L, b, args.cpu_steps, 1234)
        ncpu, blas_threads = host_threads()
        out['cpu_baseline'] = {'value': rate, 'unit': 'sweep-steps/s', 'cores': blas_threads or ncpu, 'host_cpus': ncpu,
                               'kind': 'port',
                               'sample': 'float64 NumPy oracle (einsum/BLAS, cached norm environments; B4 of BASELINE.md): 1 forward '
                                         'on the full %d-sample batch (%.2f s) + %d mid-chain sweep steps (%.1f ms each), '
                                         'extrapolated to a %d-step pass' % (b, t_fwd, args.cpu_steps, 1e3 * t_step, N - 1)}
        if args.ref_form_budget > 0 and args.config != 'c5':
            rf = cpu_baseline_reference_form(N, M, D, L, st, fcpu, y1h, args.ref_form_budget)
            out['cpu_baseline_reference_form'] = {
                'unit': 'sweep-steps/s', 'cores': 1, 'kind': 'port',
                'sample': 'the step in the reference\'s own computational form (oracle/mps_reference_form.py: broadcast-multiply-sum '
                          'contractions, float64, L2 norm environments rebuilt at every step), mid-chain, a few steps each; '
                          'forward not included',
                'forms': rf,
                'gpu_over_B3': steps_per_s / rf['B3_fixed_bond_L2']['steps_per_s'],
                'gpu_over_B1_first_sweep': steps_per_s / rf['B1_reference_truncation_L2_first_sweep']['steps_per_s']}
    drain_whole()
    if 'roofline' in out and whole['launches'] and all_pipe:
        out['roofline']['step_kernel_avg_us_hip_events_whole_run'] = 1e3 * whole['ms'] / whole['launches']
        out['roofline']['step_kernel_launches_whole_run'] = whole['launches']
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
