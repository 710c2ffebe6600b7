"""ctypes binding of libtnml_hip.so (include/tnml.h).

This is the only door to the device: there is no NumPy / CPU fallback behind it.  If the shared
library is missing or no gfx950 device is visible, the calls raise -- they never compute on the
host.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TNML_LIB', os.path.join(_HERE, 'libtnml_hip.so'))   # TNML_LIB: experiment builds

ACT = {'linear': 0, 'sigmoid': 1, 'softmax': 2}
LOSS = {'MSE': 0, 'cross_entropy': 1, 'full_cross_ent': 2}
TRUNC = {'reference': 0, 'fixed': 1, 'adaptive': 2}
SIDE_LEFT, SIDE_RIGHT = 0, 1
DBG = {'B': 0, 'dB_raw': 1, 'B_new': 2, 'sigma': 3, 'scalars': 4, 'L2_grad': 5}

# every symbol include/tnml.h declares (tests check the library exports all of them)
SYMBOLS = [
    'tnml_last_error', 'tnml_version', 'tnml_device_count', 'tnml_create', 'tnml_destroy',
    'tnml_synchronize', 'tnml_comm_unique_id', 'tnml_comm_init', 'tnml_set_cores', 'tnml_cores_size',
    'tnml_get_cores', 'tnml_scale_cores', 'tnml_set_input', 'tnml_set_labels', 'tnml_forward', 'tnml_forward_logabsmax', 'tnml_f_absmax',
    'tnml_set_f', 'tnml_get_f', 'tnml_sweep', 'tnml_activation', 'tnml_get_env', 'tnml_debug_enable',
    'tnml_get_step_debug', 'tnml_l_pos', 'tnml_batch', 'tnml_timer_start', 'tnml_timer_stop',
    'tnml_profile_enable', 'tnml_profile_get', 'tnml_profile_reset', 'tnml_svd_stats', 'tnml_trunc_rank',
    'tnml_update_B', 'tnml_l2_term', 'tnml_svd_split', 'tnml_set_svd_stop', 'tnml_set_narrow_path', 'tnml_predict', 'tnml_set_trunc_threshold',
    'tnml_set_sync_interval', 'tnml_set_step_pipeline', 'tnml_stage_batch', 'tnml_select_batch', 'tnml_get_counters',
    'tnml_svd_stats_ex', 'tnml_set_persistent', 'tnml_set_chain_path', 'tnml_marker', 'tnml_set_comm_overlap', 'tnml_comm_probe', 'tnml_set_flag_handoffs',
]


class TnmlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('tnml error %d: %s' % (code, msg))
        self.code = code


_lib = None


def lib():
    """The loaded library; raises if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C tensornetworkforml_amd/csrc).  There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.tnml_last_error.restype = C.c_char_p
        L.tnml_version.restype = C.c_char_p
        f32p, i32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        vp = C.c_void_p
        L.tnml_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.tnml_destroy.argtypes = [vp]
        L.tnml_synchronize.argtypes = [vp]
        L.tnml_comm_unique_id.argtypes = [vp]
        L.tnml_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
        L.tnml_set_cores.argtypes = [vp, f32p, C.c_size_t, i32p, C.c_int]
        L.tnml_cores_size.argtypes = [vp, C.POINTER(C.c_size_t)]
        L.tnml_get_cores.argtypes = [vp, f32p, C.c_size_t, i32p, C.POINTER(C.c_int)]
        L.tnml_scale_cores.argtypes = [vp, C.c_double]
        L.tnml_set_input.argtypes = [vp, f32p, i32p, C.c_int]
        L.tnml_set_labels.argtypes = [vp, i32p, C.c_int]
        L.tnml_forward.argtypes = [vp, f32p]
        L.tnml_f_absmax.argtypes = [vp, f64p]
        L.tnml_forward_logabsmax.argtypes = [vp, f64p]
        L.tnml_set_f.argtypes = [vp, f32p]
        L.tnml_get_f.argtypes = [vp, f32p]
        L.tnml_sweep.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                 C.c_int, C.c_float, C.c_int, f32p, f32p]
        L.tnml_activation.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_int, f32p, f32p]
        L.tnml_get_env.argtypes = [vp, C.c_int, C.c_int, f32p, C.c_size_t, C.POINTER(C.c_int)]
        L.tnml_debug_enable.argtypes = [vp, C.c_int]
        L.tnml_get_step_debug.argtypes = [vp, C.c_int, f64p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.tnml_l_pos.argtypes = [vp]
        L.tnml_batch.argtypes = [vp]
        L.tnml_timer_start.argtypes = [vp]
        L.tnml_timer_stop.argtypes = [vp, f64p]
        L.tnml_profile_enable.argtypes = [vp, C.c_int]
        L.tnml_profile_get.argtypes = [vp, C.c_int, f64p, C.POINTER(C.c_longlong)]
        L.tnml_profile_reset.argtypes = [vp]
        L.tnml_svd_stats.argtypes = [vp, C.c_int, f64p]
        L.tnml_svd_stats_ex.argtypes = [vp, C.c_int, f64p, C.c_int]
        L.tnml_set_persistent.argtypes = [vp, C.c_int]
        L.tnml_trunc_rank.argtypes = [C.c_int] * 9
        L.tnml_update_B.argtypes = [vp, f32p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float,
                                    f64p, C.c_size_t, f32p]
        L.tnml_l2_term.argtypes = [vp, f32p, C.c_int, C.c_float, f64p, f64p, C.c_size_t]
        L.tnml_set_svd_stop.argtypes = [vp, C.c_double]
        L.tnml_set_narrow_path.argtypes = [vp, C.c_int]
        L.tnml_set_chain_path.argtypes = [vp, C.c_int]
        L.tnml_marker.argtypes = [vp, C.c_int]
        L.tnml_set_comm_overlap.argtypes = [vp, C.c_int]
        L.tnml_set_flag_handoffs.argtypes = [vp, C.c_int]
        L.tnml_comm_probe.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.tnml_predict.argtypes = [vp, f32p, C.c_int, f32p]
        L.tnml_set_trunc_threshold.argtypes = [vp, C.c_double]
        L.tnml_set_sync_interval.argtypes = [vp, C.c_int]
        L.tnml_set_step_pipeline.argtypes = [vp, C.c_int]
        L.tnml_stage_batch.argtypes = [vp, C.c_int, f32p, i32p, C.c_int]
        L.tnml_select_batch.argtypes = [vp, C.c_int]
        L.tnml_get_counters.argtypes = [vp, f64p]
        L.tnml_svd_split.argtypes = [vp, f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, f64p]
        _lib = L
    return _lib


def device_count():
    return int(lib().tnml_device_count())


def _chk(rc):
    if rc != 0:
        raise TnmlError(rc, lib().tnml_last_error().decode('utf-8', 'replace'))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def trunc_rank(policy, left_dir, p, N, ml, D, mr, L, M):
    """Rank kept by tensor_svd; negative where the reference itself raises."""
    return int(lib().tnml_trunc_rank(TRUNC[policy], int(left_dir), p, N, ml, D, mr, L, M))


def comm_unique_id():
    buf = (C.c_ubyte * 128)()
    _chk(lib().tnml_comm_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf)


def core_shapes(bond, l_pos, D, L):
    N = len(bond) + 1
    shapes = []
    for i in range(N):
        ml = 1 if i == 0 else int(bond[i - 1])
        mr = 1 if i == N - 1 else int(bond[i])
        shapes.append((ml, D, mr, L) if i == l_pos else (ml, D, mr))
    return shapes


class Context:
    """One device context = one MPS + one resident batch (include/tnml.h)."""

    def __init__(self, N, D, L, M, b_capacity, device=0):
        self.N, self.D, self.L, self.M = int(N), int(D), int(L), int(M)
        self._h = C.c_void_p()
        _chk(lib().tnml_create(C.byref(self._h), self.N, self.D, self.L, self.M, int(b_capacity), int(device)))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            lib().tnml_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- multi-GPU
    def comm_init(self, rank, nranks, uid):
        buf = (C.c_ubyte * 128).from_buffer_copy(uid)
        _chk(lib().tnml_comm_init(self._h, int(rank), int(nranks), C.cast(buf, C.c_void_p)))

    # ---- parameters
    def set_cores(self, cores, l_pos):
        """cores: list of canonical arrays (ml, D, mr[, L])."""
        bond = np.array([cores[i].shape[2] for i in range(self.N - 1)], dtype=np.int32)
        flat = _f32(np.concatenate([np.asarray(c, dtype=np.float32).ravel() for c in cores]))
        _chk(lib().tnml_set_cores(self._h, _ptr(flat, C.c_float), flat.size, _ptr(bond, C.c_int32), int(l_pos)))

    def get_cores(self):
        n = C.c_size_t()
        _chk(lib().tnml_cores_size(self._h, C.byref(n)))
        flat = np.empty(n.value, dtype=np.float32)
        bond = np.empty(self.N - 1, dtype=np.int32)
        lp = C.c_int()
        _chk(lib().tnml_get_cores(self._h, _ptr(flat, C.c_float), flat.size, _ptr(bond, C.c_int32), C.byref(lp)))
        cores, off = [], 0
        for shp in core_shapes(bond, lp.value, self.D, self.L):
            k = int(np.prod(shp))
            cores.append(flat[off:off + k].reshape(shp).copy())
            off += k
        return cores, bond, lp.value

    def scale_cores(self, factor):
        _chk(lib().tnml_scale_cores(self._h, float(factor)))

    # ---- batch
    def set_input(self, X, y=None):
        X = _f32(X)
        assert X.ndim == 3 and X.shape[1] == self.N and X.shape[2] == self.D, \
            "The 1 dimension of the input data must be the flattened number of pixels"
        yp = None
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.int32)
            assert y.shape == (X.shape[0],)
            yp = _ptr(y, C.c_int32)
        _chk(lib().tnml_set_input(self._h, _ptr(X, C.c_float), yp, X.shape[0]))
        self.b = X.shape[0]

    def stage_batch(self, slot, X, y):
        """Copy a batch into device slot `slot` (0..7); `select_batch` later makes it resident without host traffic."""
        X = _f32(X)
        assert X.ndim == 3 and X.shape[1] == self.N and X.shape[2] == self.D
        y = np.ascontiguousarray(y, dtype=np.int32)
        assert y.shape == (X.shape[0],)
        _chk(lib().tnml_stage_batch(self._h, int(slot), _ptr(X, C.c_float), _ptr(y, C.c_int32), X.shape[0]))

    def select_batch(self, slot):
        _chk(lib().tnml_select_batch(self._h, int(slot)))
        self.b = int(lib().tnml_batch(self._h))

    def set_labels(self, y):
        y = np.ascontiguousarray(y, dtype=np.int32)
        _chk(lib().tnml_set_labels(self._h, _ptr(y, C.c_int32), y.shape[0]))

    # ---- hot path
    def forward(self, want_f=True):
        if not want_f:
            _chk(lib().tnml_forward(self._h, None))
            return None
        f = np.empty((self.L, self.b), dtype=np.float32)
        _chk(lib().tnml_forward(self._h, _ptr(f, C.c_float)))
        return f

    def predict(self, X):
        """f (L, b) for a batch that does not become resident (no environments are stored)."""
        X = _f32(X)
        assert X.ndim == 3 and X.shape[1] == self.N and X.shape[2] == self.D, \
            "The 1 dimension of the input data must be the flattened number of pixels"
        f = np.empty((self.L, X.shape[0]), dtype=np.float32)
        _chk(lib().tnml_predict(self._h, _ptr(X, C.c_float), X.shape[0], _ptr(f, C.c_float)))
        return f

    def forward_logabsmax(self):
        """log max|f| of the resident batch, exact even where f under/overflows float32."""
        v = C.c_double()
        _chk(lib().tnml_forward_logabsmax(self._h, C.byref(v)))
        return v.value

    def f_absmax(self):
        v = C.c_double()
        _chk(lib().tnml_f_absmax(self._h, C.byref(v)))
        return v.value

    def set_f(self, f):
        f = _f32(f)
        assert f.shape == (self.L, self.b)
        _chk(lib().tnml_set_f(self._h, _ptr(f, C.c_float)))

    def get_f(self):
        f = np.empty((self.L, self.b), dtype=np.float32)
        _chk(lib().tnml_get_f(self._h, _ptr(f, C.c_float)))
        return f

    def sweep(self, left_dir, n_steps, first_of_sweep, lr, weight_dec, L2_flag, act_fn, loss_fn, T, trunc,
              want_metrics=True, want_f=True):
        met = np.empty((n_steps, 2), dtype=np.float32) if want_metrics else None
        f = np.empty((self.L, self.b), dtype=np.float32) if want_f else None
        _chk(lib().tnml_sweep(self._h, int(bool(left_dir)), int(n_steps), int(bool(first_of_sweep)), float(lr),
                              float(weight_dec), int(bool(L2_flag)), ACT[act_fn], LOSS[loss_fn], float(T),
                              TRUNC[trunc], _ptr(met, C.c_float) if want_metrics else None,
                              _ptr(f, C.c_float) if want_f else None))
        return met, f

    def update_B(self, B, left_dir, lr, weight_dec, L2_flag, act_fn, loss_fn, T):
        """Updated merged tensor of the two sites at l_pos (canonical (ml, D, D, mr, L)) and the step's
        (accuracy, MAE); B = None uses the product of the two cores."""
        shape = None
        bp = None
        if B is not None:
            B = _f32(B)
            shape = B.shape
            bp = _ptr(B, C.c_float)
        cap = 4 * max(self.M, self.D * self.L) ** 2 * self.D * self.D * self.L
        out = np.empty(cap, dtype=np.float64)
        met = np.empty(2, dtype=np.float32)
        _chk(lib().tnml_update_B(self._h, bp, int(bool(left_dir)), float(lr), float(weight_dec), int(bool(L2_flag)),
                                 ACT[act_fn], LOSS[loss_fn], float(T), _ptr(out, C.c_double), out.size,
                                 _ptr(met, C.c_float)))
        return (out[:int(np.prod(shape))].reshape(shape).copy() if shape is not None else out), met

    def l2_term(self, B, left_dir, weight_dec):
        """(wd <B, Ln.B.Rn>, 2 wd Ln.B.Rn) for a merged tensor B (canonical layout) at l_pos."""
        B = _f32(B)
        grad = np.empty(B.size, dtype=np.float64)
        loss = C.c_double()
        _chk(lib().tnml_l2_term(self._h, _ptr(B, C.c_float), int(bool(left_dir)), float(weight_dec), C.byref(loss),
                                _ptr(grad, C.c_double), grad.size))
        return loss.value, grad.reshape(B.shape)

    def svd_split(self, mat, m):
        """(U sqrt(S) [rows, m], sqrt(S) Vh [m, cols], all singular values) of a 2-D matrix."""
        mat = _f32(mat)
        rows, cols = mat.shape
        US = np.empty((rows, int(m)), dtype=np.float32)
        SVh = np.empty((int(m), cols), dtype=np.float32)
        sig = np.empty(min(rows, cols), dtype=np.float64)
        _chk(lib().tnml_svd_split(self._h, _ptr(mat, C.c_float), rows, cols, int(m), _ptr(US, C.c_float),
                                  _ptr(SVh, C.c_float), _ptr(sig, C.c_double)))
        return US, SVh, sig

    def activation(self, act_fn, loss_fn, T, want_act=True, want_der=False, input_is_activated=False):
        a = np.empty((self.L, self.b), dtype=np.float32) if want_act else None
        d = np.empty((self.L, self.b), dtype=np.float32) if want_der else None
        _chk(lib().tnml_activation(self._h, ACT[act_fn], LOSS[loss_fn], float(T), int(bool(input_is_activated)),
                                   _ptr(a, C.c_float) if want_act else None,
                                   _ptr(d, C.c_float) if want_der else None))
        return a, d

    def loss_derivative_of_activated(self, act_fn, loss_fn, T):
        return self.activation(act_fn, loss_fn, T, want_act=False, want_der=True, input_is_activated=True)[1]

    # ---- inspection
    def get_env(self, side, site):
        out = np.empty(self.b * max(self.M, self.D * self.L) * 2, dtype=np.float32)
        m = C.c_int()
        _chk(lib().tnml_get_env(self._h, int(side), int(site), _ptr(out, C.c_float), out.size, C.byref(m)))
        return out[:self.b * m.value].reshape(self.b, m.value).copy()

    def set_svd_stop(self, stop2):
        """Jacobi stopping threshold on g^2 / scale^2 (include/tnml.h); default 1e-6."""
        _chk(lib().tnml_set_svd_stop(self._h, float(stop2)))

    def set_trunc_threshold(self, threshold):
        """Cumulative-share threshold of trunc='adaptive' (default 0.999)."""
        _chk(lib().tnml_set_trunc_threshold(self._h, float(threshold)))

    def set_step_pipeline(self, on):
        """True (default): one launch per sweep step (pipelined); False: the classic launch sequence."""
        _chk(lib().tnml_set_step_pipeline(self._h, int(on)))

    def set_persistent(self, on=True):
        """1 / True (default): a full sweep as ONE persistent launch; 2: one launch per role on three streams; 0 / False: one launch per step."""
        _chk(lib().tnml_set_persistent(self._h, int(on)))

    def set_sync_interval(self, n_steps):
        """Drain the stream every n_steps sweep steps (0: never); for runs under a dispatch-intercepting profiler."""
        _chk(lib().tnml_set_sync_interval(self._h, int(n_steps)))

    def set_comm_overlap(self, on=True):
        """Communicator path: update side and batch side + all-reduce on two streams (default) or one fused launch per step."""
        _chk(lib().tnml_set_comm_overlap(self._h, int(bool(on))))

    def comm_probe(self, n_floats, reps=200):
        """Mean device time (us) of one all-reduce of n_floats floats on the exchange stream (collective call)."""
        v = C.c_double()
        _chk(lib().tnml_comm_probe(self._h, int(n_floats), int(reps), C.byref(v)))
        return v.value

    def set_flag_handoffs(self, on=True):
        """hand-offs between the context's two streams as sequence numbers in memory (default) or as events (tools that
        serialise dispatches: rocprofv3 --pmc)"""
        _chk(lib().tnml_set_flag_handoffs(self._h, int(bool(on))))

    def set_chain_path(self, force_plain):
        """Forward chain as plain FMAs (True) instead of the matrix-core kernel (tests, diagnostics)."""
        _chk(lib().tnml_set_chain_path(self._h, int(bool(force_plain))))

    def set_narrow_path(self, force_large):
        """True: every step takes the large-tensor (HBM-resident) path; False: automatic."""
        _chk(lib().tnml_set_narrow_path(self._h, int(bool(force_large))))

    def debug_enable(self, on=True):
        """True / 1: capture every step's tensors; 2: cycle stamps only; 4: check every launch; False / 0: off."""
        _chk(lib().tnml_debug_enable(self._h, int(on)))

    def step_debug(self, what):
        cap = max(4 * max(self.M, self.D * self.L) ** 2 * self.D * self.D * self.L + 64, 512)
        out = np.empty(cap, dtype=np.float64)
        n = C.c_size_t()
        _chk(lib().tnml_get_step_debug(self._h, DBG[what], _ptr(out, C.c_double), out.size, C.byref(n)))
        return out[:n.value].copy()

    @property
    def l_pos(self):
        return int(lib().tnml_l_pos(self._h))

    def synchronize(self):
        _chk(lib().tnml_synchronize(self._h))

    # ---- measurement
    def marker(self, marker_id):
        """Phase boundary for profilers: an empty kernel of `marker_id` workgroups on the context's stream."""
        _chk(lib().tnml_marker(self._h, int(marker_id)))

    def timer_start(self):
        _chk(lib().tnml_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_double()
        _chk(lib().tnml_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        """True / 1: events around every launch (synchronising); 2: one event pair per sweep call; False / 0: off."""
        _chk(lib().tnml_profile_enable(self._h, int(on)))

    def profile_reset(self):
        _chk(lib().tnml_profile_reset(self._h))

    def svd_stats(self, reset=False):
        """(total Jacobi sweeps, SVDs, total rounds) since the last reset; `cholesky_steps` holds the fourth counter."""
        out = (C.c_double * 4)()
        _chk(lib().tnml_svd_stats_ex(self._h, int(bool(reset)), out, 4))
        self.cholesky_steps = out[3]
        return out[0], out[1], out[2]

    def counters(self):
        """Work since the last profile_reset: dict of sweep steps, algorithmic bytes / flops, forwards, launches, device ms."""
        out = (C.c_double * 8)()
        _chk(lib().tnml_get_counters(self._h, out))
        keys = ('sweep_steps', 'algorithmic_bytes', 'algorithmic_flops', 'forwards', 'forward_bytes', 'launches', 'pipelined_steps', 'sweep_device_ms')
        return dict(zip(keys, [float(v) for v in out]))

    def profile_get(self, which):
        ms, n = C.c_double(), C.c_longlong()
        _chk(lib().tnml_profile_get(self._h, int(which), C.byref(ms), C.byref(n)))
        return ms.value, n.value
