"""Host-side engine: turns PyTorch-ROCm tensors into raw pointers for the C ABI.

One instance per process.  It owns no device memory: every buffer (state, E, Vt, ...)
is a torch tensor allocated by the caller's caching allocator on the caller's current
stream, which is also the stream the kernels are enqueued on -- so ordering with the
surrounding PyTorch ops needs no extra synchronisation.
"""
import torch

from . import _lib

NW, SW = _lib.SDP_NW, _lib.SDP_SW


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


def _ptr(t):
    return None if t is None else t.data_ptr()


EXACT_STATE = 0x100  # include/sdp.h: SDP_EXACT_STATE
ET_BROADCAST = 0x200  # include/sdp.h: SDP_ET_BROADCAST
REF_ROUNDING = 0x400  # include/sdp.h: SDP_REF_ROUNDING
REF = "ref"           # value of `exact_state` that selects it: the state is then the reference's own (B, N, M, 3) fp32
F64 = "f64"           # value of `exact_state` that rides along with float64 tensors: the state is (B, N, M, 3) float64
TRACEBACK_RULES = {"cpu": 0, "cuda": 1}  # include/sdp.h: SDP_TRACEBACK_CPU / SDP_TRACEBACK_CUDA


class HipEngine:
    """Thin veneer over libsdp_hip.so.  All tensors must be fp32, contiguous, on one ROCm device."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        # optional profiling hook (bench.py): callable(name) -> context manager that brackets one
        # kernel launch on the current stream, e.g. with a pair of events.  None = no overhead.
        self.launch_hook = None
        # per-pass wave-count override for experiments and tests ({0 fwd, 1 bwd, 2 adj-fwd, 3 adj-bwd} -> waves);
        # travels with each call as SDP_WAVES(w), the library keeps no tuning state
        self.force_waves = {}
        # measurement control (bench.py's `no_skip` figures): False = the backward sweep runs every chunk (SDP_NO_ZERO_SKIP);
        # results are bit-identical either way
        self.zero_skip = True
        self._labels = {}

    # kernel ids of sdp_plan (csrc/sdp_api.hip: variant()) -> the symbol rocprofv3 will show for the launch
    KERNEL_NAMES = {0: "sdp_fwd_kernel", 1: "sdp_bwd_kernel", 2: "sdp_adj_fwd_kernel", 3: "sdp_adj_bwd_kernel", 4: "sdp_bwd_lat_kernel",
                    5: "sdp_fwd_x_kernel", 6: "sdp_fwd_lat_kernel", 7: "sdp_bwd_x_kernel", 8: "sdp_bwd_x_lat_kernel", 9: "sdp_fwd_x_tp_kernel",
                    10: "sdp_adj_fwd_loss_kernel", 11: "sdp_fwd_g_kernel", 12: "sdp_bwd_g_kernel", 14: "sdp_adj_bwd_g_kernel",
                    15: "sdp_bwd_lat_g_kernel", 18: "sdp_bwd_x_g_kernel", 19: "sdp_bwd_x_lat_g_kernel", 20: "sdp_fwd_x_tp_g_kernel",
                    21: "sdp_fwd_p_kernel", 22: "sdp_fwd_x_tp_p_kernel", 23: "sdp_bwd_p_kernel", 24: "sdp_bwd_x_p_kernel",
                    25: "sdp_fwd_pg_kernel", 26: "sdp_fwd_x_tp_pg_kernel", 27: "sdp_bwd_pg_kernel", 28: "sdp_bwd_x_pg_kernel",
                    29: "sdp_fwd18_kernel", 30: "sdp_fwd18_lat_kernel", 31: "sdp_fwd18_g_kernel", 32: "sdp_bwd18_kernel",
                    33: "sdp_bwd18_lat_kernel", 34: "sdp_bwd18_g_kernel", 35: "sdp_bwd18_lat_g_kernel", 36: "sdp_bwd_pipe_kernel"}

    def _label(self, pass_, B, N, M, has_lens, exact, dev, default):
        """Name of the kernel a launch will use (for the launch hook: bench.py's per-kernel timers must carry the names the
        rocprofv3 summaries carry).  Asked of the library's own launch policy (sdp_plan), once per problem."""
        if self.launch_hook is None:
            return default
        key = (pass_, B, N, M, bool(has_lens), bool(exact), dev)
        got = self._labels.get(key)
        if got is None:
            import ctypes
            kid = ctypes.c_int(-1)
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            rc = self.lib.sdp_plan(pass_, B, N, M, 1 if has_lens else 0, 1 if exact else 0, cus, ctypes.byref(kid), None, None, None)
            got = self.KERNEL_NAMES.get(kid.value, default) if rc == 0 else default
            self._labels[key] = got
        return got

    def _v(self, pass_, variant):
        w = self.force_waves.get(pass_, 0)
        return variant | (_lib.SDP_WAVES(w) if w else 0)

    def check_device(self, device=None):
        """Raise HandoffTimeout if a kernel launched earlier on `device` reported a strip hand-off that timed out.
        Host-side read: synchronise first to cover work that is still in flight."""
        import ctypes
        dev = torch.cuda.current_device() if device is None else device
        info = (ctypes.c_int32 * 4)()
        _lib.check(self.lib.sdp_device_status(dev, info), "sdp_device_status")
        return list(info)

    def _bracket(self, name):
        return self.launch_hook(name) if self.launch_hook is not None else _NULL_CTX

    # ---- helpers -------------------------------------------------------------------
    @staticmethod
    def _dev(t):
        if not t.is_cuda:
            raise RuntimeError(
                "deepblast_amd runs on a ROCm device only (got a CPU tensor); there is no CPU fallback. "
                "Move theta/A to 'cuda'.")
        return t.device.index if t.device.index is not None else torch.cuda.current_device()

    @staticmethod
    def _check(ref, **tensors):
        """The C ABI takes raw addresses: refuse anything that is not fp32 on the launch device."""
        for name, t in tensors.items():
            if t is None:
                continue
            if t.dtype != torch.float32:
                raise TypeError(f"{name} must be torch.float32, got {t.dtype}")
            if t.device != ref.device:
                raise ValueError(f"{name} is on {t.device}, expected {ref.device}")

    @staticmethod
    def _stream(dev):
        return torch.cuda.current_stream(dev).cuda_stream

    def max_cols(self):
        return self.lib.sdp_max_cols()

    def new_state(self, B, N, M, device, derivative=False, ref=False):
        """Opaque buffer for Q (packed: two 20-bit weights, 5 bytes per cell) or, with derivative=True, for Qd (float2 per cell); ref: the
        reference-rounding mode's (B, N, M, 3) fp32 for either."""
        if ref:
            nbytes = self.lib.sdp_state_bytes_v(B, N, M, REF_ROUNDING)
        else:
            nbytes = (self.lib.sdp_state_d_bytes if derivative else self.lib.sdp_state_bytes)(B, N, M)
        return torch.empty(nbytes // 4, dtype=torch.float32, device=device)

    @staticmethod
    def _state_flags(exact_state):
        """`exact_state` of forward / backward -> flag bits: False = packed, True = float2, "ref" = reference rounding."""
        if isinstance(exact_state, str):
            if exact_state != REF:
                raise ValueError(f"exact_state must be False, True or {REF!r}, got {exact_state!r}")
            return REF_ROUNDING
        return EXACT_STATE if exact_state else 0

    @staticmethod
    def _lens(lens, B, device):
        if lens is None:
            return None
        lens = torch.as_tensor(lens, dtype=torch.int32, device=device).contiguous()
        if lens.shape != (B, 2):
            raise ValueError(f"lengths must have shape ({B}, 2), got {tuple(lens.shape)}")
        return lens

    # ---- the four passes -----------------------------------------------------------
    def forward(self, theta, A, variant, lens=None, exact_state=False):
        """-> (Vt (B,), state).  Replaces _forward_pass_kernel (nw_cuda.py:74-79).

        exact_state=False: the compact state the backward sweep reads; True: Q as float2, which the two
        adjoint sweeps need (include/sdp.h, SDP_EXACT_STATE); "ref": the reference's arithmetic and its own
        (B,N,M,3) fp32 Q (SDP_REF_ROUNDING) -- the other three sweeps must then be asked for the same."""
        dev = self._dev(theta)
        if theta.dtype == torch.float64:
            return self._forward_f64(theta, A, variant, lens, dev)
        self._check(theta, theta=theta, A=A)
        theta, A = theta.contiguous(), A.contiguous()
        B, N, M = theta.shape
        lens = self._lens(lens, B, theta.device)
        state = self.new_state(B, N, M, theta.device, derivative=bool(exact_state), ref=exact_state == REF)
        variant = variant | self._state_flags(exact_state)
        Vt = torch.empty(B, dtype=torch.float32, device=theta.device)
        # (the general-pitch builds are chosen from the pointers' alignment inside the library: the label then names the aligned twin)
        with torch.cuda.device(dev), self._bracket(self._label(0, B, N, M, lens is not None, exact_state is True, dev, "sdp_fwd_kernel")):
            rc = self.lib.sdp_forward_f32(_ptr(theta), _ptr(A), _ptr(state), _ptr(Vt), B, N, M, _ptr(lens),
                                          self._v(0, variant), dev, self._stream(dev))
        _lib.check(rc, "sdp_forward_f32")
        return Vt, state

    def state_pair_bytes(self, N, M, exact_state=False):
        """Bytes between the records of consecutive pairs in the state buffer (include/sdp.h: sdp_state_pair_stride)."""
        return self.lib.sdp_state_pair_stride(N, M, 1 if exact_state else 0)

    def backward(self, Et, state, shape, variant, lens=None, exact_state=False, pair_range=None, out=None, no_fill=False):
        """-> E (B,N,M).  Replaces _backward_pass_kernel (nw_cuda.py:98-102).

        exact_state: `state` came from forward(..., exact_state=True).
        pair_range=(lo, hi), out=(B,N,M) tensor: sweep only pairs lo..hi-1 of the batch, writing out[lo:hi] (the
        other rows of `out` are not touched) -- the backward sweep of a batch in pieces, so that a collective on
        piece k can run under the sweep of piece k+1 (distributed.py).  Needs lens=None.  The library finds the
        pairs' records in `state` itself (sdp_backward_range_f32 takes the whole batch's buffers and the range).
        no_fill (with lens): E outside each pair's block is NOT written (SDP_NO_FILL) -- for consumers that mask by the
        same lengths and never read it; the default zero-fills, as the public contract says."""
        dev = self._dev(state)
        B, N, M = shape
        if Et.device != state.device:
            raise ValueError(f"Et is on {Et.device}, expected {state.device}")
        if state.dtype == torch.float64:
            if pair_range is not None or out is not None:
                raise ValueError("the float64 path sweeps whole batches (no pair_range / out)")
            return self._backward_f64(Et, state, (B, N, M), variant, lens, dev)
        Et, bcast = self._et(Et, B)
        lens = self._lens(lens, B, state.device)
        E = torch.empty((B, N, M), dtype=torch.float32, device=state.device) if out is None else out
        if tuple(E.shape) != (B, N, M) or E.dtype != torch.float32 or not E.is_contiguous() or E.device != state.device:
            raise ValueError("out must be a contiguous float32 (B, N, M) tensor on the state's device")
        v = self._v(1, variant) | self._state_flags(exact_state) | (ET_BROADCAST if bcast else 0)
        v |= (0 if self.zero_skip else _lib.SDP_NO_ZERO_SKIP) | (_lib.SDP_NO_FILL if (no_fill and lens is not None) else 0)
        with torch.cuda.device(dev), self._bracket(self._label(1, B, N, M, lens is not None, exact_state is True, dev, "sdp_bwd_kernel")):
            if pair_range is None:
                rc = self.lib.sdp_backward_f32(_ptr(Et), _ptr(state), _ptr(E), B, N, M, _ptr(lens), v, dev, self._stream(dev))
            else:
                lo, hi = pair_range
                if lens is not None:
                    raise ValueError("pair_range needs lens=None")
                if not (0 <= lo < hi <= B):
                    raise ValueError(f"pair_range {pair_range} outside the batch of {B}")
                rc = self.lib.sdp_backward_range_f32(_ptr(Et), _ptr(state), _ptr(E), B, N, M, lo, hi - lo, v, dev, self._stream(dev))
        _lib.check(rc, "sdp_backward_f32")
        return E

    @staticmethod
    def _et(Et, B):
        """-> (tensor whose data_ptr the kernel reads, broadcast flag).  The usual (B,) fp32 contiguous cotangent goes as
        it is.  A broadcast scalar -- what `Vt.sum().backward()` hands over: a stride-0 expand of one element -- goes as
        that one element with SDP_ET_BROADCAST instead of being expanded into B floats by a kernel of its own."""
        if Et.dtype == torch.float32 and Et.dim() <= 1:
            if Et.numel() == 1 or (Et.shape == (B,) and Et.stride(0) == 0):
                return Et, True
            if Et.shape == (B,) and Et.is_contiguous():
                return Et, False
        return Et.to(torch.float32).expand(B).contiguous(), False

    def adjoint_forward(self, state, Ztheta, ZA, variant, lens=None, ref=False):
        """-> (Vtd (B,), state_d).  Replaces _adjoint_forward_pass_kernel (nw_cuda.py:134-139).
        ref: `state` came from forward(..., exact_state="ref"); the Hessian product is rounded as numpy rounds it."""
        dev = self._dev(state)
        for name, t in (("Ztheta", Ztheta), ("ZA", ZA)):
            if t is not None and t.device != state.device:
                raise ValueError(f"{name} is on {t.device}, expected {state.device}")
        if state.dtype == torch.float64:
            return self._adjoint_forward_f64(state, Ztheta, ZA, variant, lens, dev)
        Ztheta = Ztheta.to(torch.float32).contiguous()
        B, N, M = Ztheta.shape
        if ZA is not None:
            ZA = ZA.to(torch.float32).contiguous()
        lens = self._lens(lens, B, state.device)
        state_d = self.new_state(B, N, M, state.device, derivative=True, ref=ref)
        Vtd = torch.empty(B, dtype=torch.float32, device=state.device)
        with torch.cuda.device(dev), self._bracket("sdp_adj_fwd_kernel"):
            rc = self.lib.sdp_adjoint_forward_f32(_ptr(state), _ptr(Ztheta), _ptr(ZA), _ptr(Vtd), _ptr(state_d),
                                                  B, N, M, _ptr(lens), self._v(2, variant) | (REF_ROUNDING if ref else 0), dev, self._stream(dev))
        _lib.check(rc, "sdp_adjoint_forward_f32")
        return Vtd, state_d

    def adjoint_forward_loss(self, state, ref, pred, G, scale, kind, variant, lens=None):
        """Adjoint forward sweep seeded with scale[b] * d(loss term)/d(pred) formed in the kernel (include/sdp.h:
        sdp_adjoint_forward_loss_f32).  -> (Vtd (B,), state_d)."""
        dev = self._dev(state)
        self._check(state, first=ref, pred=pred, G=G, scale=scale)
        ref, pred, G, scale = ref.contiguous(), pred.contiguous(), G.contiguous(), scale.contiguous()
        B, N, M = pred.shape
        lens = self._lens(lens, B, state.device)
        state_d = self.new_state(B, N, M, state.device, derivative=True)
        Vtd = torch.empty(B, dtype=torch.float32, device=state.device)
        with torch.cuda.device(dev), self._bracket("sdp_adj_fwd_kernel"):
            rc = self.lib.sdp_adjoint_forward_loss_f32(_ptr(state), _ptr(ref), _ptr(pred), _ptr(G), _ptr(scale), kind, _ptr(Vtd),
                                                       _ptr(state_d), B, N, M, _ptr(lens), self._v(2, variant), dev,
                                                       self._stream(dev))
        _lib.check(rc, "sdp_adjoint_forward_loss_f32")
        return Vtd, state_d

    def adjoint_backward(self, E, state, state_d, variant, lens=None, ref=False):
        """-> Ed (B,N,M).  Replaces _adjoint_backward_pass_kernel (nw_cuda.py:160-165).  ref: as for adjoint_forward."""
        dev = self._dev(state)
        if state.dtype == torch.float64:
            return self._adjoint_backward_f64(E, state, state_d, variant, lens, dev)
        self._check(state, E=E, state_d=state_d)
        E = E.contiguous()
        B, N, M = E.shape
        lens = self._lens(lens, B, state.device)
        Ed = torch.empty((B, N, M), dtype=torch.float32, device=state.device)
        with torch.cuda.device(dev), self._bracket("sdp_adj_bwd_kernel"):
            rc = self.lib.sdp_adjoint_backward_f32(_ptr(E), _ptr(state), _ptr(state_d), _ptr(Ed), B, N, M,
                                                   _ptr(lens), self._v(3, variant) | (REF_ROUNDING if ref else 0) | (0 if self.zero_skip else _lib.SDP_NO_ZERO_SKIP),
                                                   dev, self._stream(dev))
        _lib.check(rc, "sdp_adjoint_backward_f32")
        return Ed

    # ---- float64 tensors (include/sdp.h: sdp_*_f64) --------------------------------------
    # The reference's CPU classes take float64 as it comes (its tests: decoding, gradcheck, gradgradcheck on .double()
    # tensors, deepblast/tests/test_nw.py:46-90).  Here: the reference-arithmetic kernels with float64 storage, one
    # workgroup per pair -- for tests and small problems, not a second fast path.  The state is the reference's own
    # (B, N, M, 3) weights in float64, and the other three sweeps recognise it by its dtype.
    @staticmethod
    def _check64(ref, **tensors):
        for name, t in tensors.items():
            if t is None:
                continue
            if t.dtype != torch.float64:
                raise TypeError(f"{name} must be torch.float64 like the other tensors of this call, got {t.dtype}")
            if t.device != ref.device:
                raise ValueError(f"{name} is on {t.device}, expected {ref.device}")

    def _forward_f64(self, theta, A, variant, lens, dev):
        self._check64(theta, theta=theta, A=A)
        theta, A = theta.contiguous(), A.contiguous()
        B, N, M = theta.shape
        lens = self._lens(lens, B, theta.device)
        state = torch.empty((B, N, M, 3), dtype=torch.float64, device=theta.device)
        Vt = torch.empty(B, dtype=torch.float64, device=theta.device)
        with torch.cuda.device(dev), self._bracket("sdp_f64_fwd_kernel"):
            rc = self.lib.sdp_forward_f64(_ptr(theta), _ptr(A), _ptr(state), _ptr(Vt), B, N, M, _ptr(lens), variant, dev, self._stream(dev))
        _lib.check(rc, "sdp_forward_f64")
        return Vt, state

    def _backward_f64(self, Et, state, shape, variant, lens, dev):
        B, N, M = shape
        Et = Et.to(torch.float64)
        bcast = Et.numel() == 1
        if not bcast:
            Et = Et.expand(B).contiguous()
        lens = self._lens(lens, B, state.device)
        E = torch.empty((B, N, M), dtype=torch.float64, device=state.device)
        with torch.cuda.device(dev), self._bracket("sdp_f64_bwd_kernel"):
            rc = self.lib.sdp_backward_f64(_ptr(Et), _ptr(state), _ptr(E), B, N, M, _ptr(lens), variant | (ET_BROADCAST if bcast else 0),
                                           dev, self._stream(dev))
        _lib.check(rc, "sdp_backward_f64")
        return E

    def _adjoint_forward_f64(self, state, Ztheta, ZA, variant, lens, dev):
        Ztheta = Ztheta.to(torch.float64).contiguous()
        B, N, M = Ztheta.shape
        if ZA is not None:
            ZA = ZA.to(torch.float64).contiguous()
        lens = self._lens(lens, B, state.device)
        state_d = torch.empty((B, N, M, 3), dtype=torch.float64, device=state.device)
        Vtd = torch.empty(B, dtype=torch.float64, device=state.device)
        with torch.cuda.device(dev), self._bracket("sdp_f64_adj_fwd_kernel"):
            rc = self.lib.sdp_adjoint_forward_f64(_ptr(state), _ptr(Ztheta), _ptr(ZA), _ptr(Vtd), _ptr(state_d), B, N, M, _ptr(lens),
                                                  variant, dev, self._stream(dev))
        _lib.check(rc, "sdp_adjoint_forward_f64")
        return Vtd, state_d

    def _adjoint_backward_f64(self, E, state, state_d, variant, lens, dev):
        self._check64(state, E=E, state_d=state_d)
        E = E.contiguous()
        B, N, M = E.shape
        lens = self._lens(lens, B, state.device)
        Ed = torch.empty((B, N, M), dtype=torch.float64, device=state.device)
        with torch.cuda.device(dev), self._bracket("sdp_f64_adj_bwd_kernel"):
            rc = self.lib.sdp_adjoint_backward_f64(_ptr(E), _ptr(state), _ptr(state_d), _ptr(Ed), B, N, M, _ptr(lens), variant, dev,
                                                   self._stream(dev))
        _lib.check(rc, "sdp_adjoint_backward_f64")
        return Ed

    def traceback(self, grad, lens=None, rule="cpu"):
        """Batched traceback on the device -> (states (B,cap,3) int32, counts (B,) int32).

        rule: "cpu" = the CPU classes' walk (nw.py:401-444), "cuda" = the walk of the GPU classes this library
        replaces (nw_cuda.py:273-317: stops as soon as one neighbour is off the matrix)."""
        if rule not in TRACEBACK_RULES:
            raise ValueError(f"traceback rule must be one of {sorted(TRACEBACK_RULES)}, got {rule!r}")
        dev = self._dev(grad)
        grad = grad.detach().to(torch.float32).contiguous()
        B, N, M = grad.shape
        lens = self._lens(lens, B, grad.device)
        cap = self.lib.sdp_traceback_capacity(N, M)
        states = torch.empty((B, cap, 3), dtype=torch.int32, device=grad.device)
        counts = torch.empty(B, dtype=torch.int32, device=grad.device)
        with torch.cuda.device(dev), self._bracket("sdp_traceback_kernel"):
            rc = self.lib.sdp_traceback_rule_i32(_ptr(grad), _ptr(states), _ptr(counts), B, N, M, _ptr(lens),
                                                 TRACEBACK_RULES[rule], dev, self._stream(dev))
        _lib.check(rc, "sdp_traceback_rule_i32")
        return states, counts

    def init(self, device=None):
        """Create the library's per-device state now (include/sdp.h: sdp_init) -- needed only before stream capture."""
        dev = torch.cuda.current_device() if device is None else device
        _lib.check(self.lib.sdp_init(dev), "sdp_init")

    def selftest(self, device=0):
        _lib.check(self.lib.sdp_selftest(device), "sdp_selftest")


_ENGINE = None


def get_engine():
    """The process-wide engine.  Raises (ImportError) if the HIP library is not built."""
    global _ENGINE
    if _ENGINE is None:
        _ENGINE = HipEngine()
    return _ENGINE
