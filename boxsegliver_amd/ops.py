"""Host-side operators of the U-Net hot path: thin wrappers over the libunetk C ABI plus the
torch.autograd.Function nodes the networks are composed from.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every FLOP and every
byte moved on the hot path happens inside a hand-written HIP kernel reached through
boxsegliver_amd._abi.  Nothing in this module falls back to torch ops or to the CPU oracle.

Reference call sites replaced (relative to the reference repo root):
  Conv3x3BnRelu : slim.repeat(x, 2, slim.conv2d, C, 3) unit           NetworksV2/UNet.py:79,85,94
  MaxPool2x2    : slim.max_pool2d(x, [2, 2])                           NetworksV2/UNet.py:81
  DeconvConcat  : slim.conv2d_transpose(x, C/2, 2, 2) + tf.concat      NetworksV2/UNet.py:91-93
  HeadLoss      : logits 1x1 conv + loss + metrics                     UNet.py:97-155, loss_metrics.py:115-339
"""
import ctypes
import os

import torch

from . import _abi
from ._abi import Conv3dDesc, ConvDesc, Deconv3dDesc, DeconvDesc, HeadDesc, NormDesc, check, ptr, stream_ptr


# ----------------------------------------------------------------------------- memory helpers
class _Workspace(object):
    """One growing scratch buffer per device; ops on one stream use it back to back."""

    def __init__(self):
        self._bufs = {}

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 16)
        key = (device.type, device.index)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf


WORKSPACE = _Workspace()
FUSE_EVAL = os.environ.get("UNETK_FUSE_EVAL", "1") != "0"    # inference: conv + (scale, shift) + ReLU [+ pool] in one kernel (0: two passes, measurement)
FUSE_NBR = True         # conv2's input gradient also emits conv1's norm-backward reduction (unetk_conv3x3_dgrad_nbr)
FUSED_NBR = {}          # dx.data_ptr() -> (producer y.data_ptr(), shape, partials, rows, dx._version); consumed by the producer's backward
DEBUG_CAPTURE = None    # tools/: set to a list to record each conv unit's backward operands
PROFILE_SHAPES = False  # bench.py --detail: one row per (kernel, layer shape)
PROFILE = None          # bench.py (profile_on): a list -> (op tag, algorithmic FLOPs, first trace record, end record, algorithmic bytes)


# ----------------------------------------------------------------------------- packed-filter cache
PACK_BATCH = True       # re-pack every known filter in ONE launch (unetk_pack_many) after the variables changed
PARAM_GEN = 0           # bumped by the kernels that write variables behind torch's back (optimiser steps, broadcasts)
_PARAM_FLATS = {}       # storage data_ptr of a ParamStore's flat buffer -> weakref to that tensor


def register_param_buffer(flat):
    """ParamStore: filters that are views of `flat` may keep their packed forms between steps."""
    import weakref
    _PARAM_FLATS[flat.untyped_storage().data_ptr()] = weakref.ref(flat)


def bump_param_gen():
    global PARAM_GEN
    PARAM_GEN += 1


class _PackItem(ctypes.Structure):            # unetk_pack_item
    _fields_ = [("kind", ctypes.c_int32), ("perm", ctypes.c_int32), ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32),
                ("block0", ctypes.c_int32), ("nblocks", ctypes.c_int32), ("r0", ctypes.c_int32), ("r1", ctypes.c_int32),
                ("w", ctypes.c_void_p), ("wp_fwd", ctypes.c_void_p), ("wp_dgrad", ctypes.c_void_p), ("r2", ctypes.c_void_p)]


class _PackCache(object):
    """Packed filters of the variables, kept between calls and rebuilt together.  An entry is valid while neither torch
    (the flat buffer's version counter) nor the optimiser kernels (PARAM_GEN) have written the variables since it was
    packed.  Only views of a registered ParamStore buffer are cached; anything else is packed on the spot."""

    def __init__(self):
        self.entries = {}          # key -> dict(flat=weakref, gen, wp_f, wp_d, items=[(kind, perm, cin, cout, w_ptr, f_ptr, d_ptr)])
        self.table = None          # (device tensor, n_items, total_blocks) of ALL live entries
        self.hits = self.batched = self.singles = 0

    @staticmethod
    def _flat_of(w):
        ref = _PARAM_FLATS.get(w.untyped_storage().data_ptr())
        return ref() if ref is not None else None

    def get(self, w, key, build):
        """build() -> (wp_f, wp_d, items): packs this filter alone.  Returns (wp_f, wp_d)."""
        flat = self._flat_of(w) if PACK_BATCH else None
        if flat is None:
            wp_f, wp_d, _ = build()
            return wp_f, wp_d
        key = (w.data_ptr(), tuple(w.shape)) + tuple(key)
        gen = (PARAM_GEN, flat._version)
        ent = self.entries.get(key)
        if ent is not None and ent["flat"]() is flat:
            if ent["gen"] == gen:
                self.hits += 1
                return ent["wp_f"], ent["wp_d"]
            self._repack_all()
            if ent["gen"] == gen:
                return ent["wp_f"], ent["wp_d"]
        import weakref
        for k in [k for k, e in self.entries.items() if e["flat"]() is None]:      # variables of models that are gone
            del self.entries[k]
            self.table = None
        wp_f, wp_d, items = build()
        self.singles += 1
        self.entries[key] = dict(flat=weakref.ref(flat), gen=gen, wp_f=wp_f, wp_d=wp_d, items=items)
        self.table = None
        return wp_f, wp_d

    def _repack_all(self):
        # strong references for the duration of the call: a collection cycle in the middle of it (a freed model's flat buffer)
        # must not turn a weak reference dead between this check and its use below
        held = {k: e["flat"]() for k, e in self.entries.items()}
        dead = [k for k, f in held.items() if f is None]
        for k in dead:
            del self.entries[k]
            self.table = None
        live = list(self.entries.values())
        if not live:
            return
        dev = live[0]["wp_f"].device
        if self.table is None:
            rows, block0 = [], 0
            for e in live:
                for kind, perm, cin, cout, w_ptr, f_ptr, d_ptr in e["items"]:
                    nb = _abi.lib().unetk_pack_item_blocks(kind, cin, cout)
                    if nb <= 0:
                        check(nb, "pack_item_blocks")
                    rows.append(_PackItem(kind, perm, cin, cout, block0, nb, 0, 0, w_ptr, f_ptr, d_ptr, None))
                    block0 += nb
            arr = (_PackItem * len(rows))(*rows)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.table = (host.to(dev), len(rows), block0)
        tab, n_items, total = self.table
        check(_abi.lib().unetk_pack_many(ptr(tab), n_items, total, stream_ptr()), "pack_many")
        self.batched += 1
        for k, e in self.entries.items():
            e["gen"] = (PARAM_GEN, held[k]._version)


PACKS = _PackCache()
PK_CONV_F32, PK_CONV_BF16, PK_DECONV_F32, PK_DECONV_BF16 = 0, 1, 2, 3


class _Timed(object):
    """Brackets one C-ABI call with two marks of the library's kernel trace (csrc/prof.hip): while bench.py has the trace on,
    every kernel the call launches carries start / stop events bound to its dispatch, and the bracket is the half-open range of
    trace records [i0, i1) -- its time is the sum of those kernels' own GPU durations (what rocprofv3 --kernel-trace reports),
    with no host gap inside whatever the stream's queue depth was."""

    def __init__(self, tag, flops, shape=None, nbytes=0):
        """flops: algorithmic FLOPs of a matrix kernel; nbytes: algorithmic HBM bytes (each operand once) of an HBM-bound pass."""
        self.on = PROFILE is not None
        self.tag, self.flops, self.nbytes = (tag if not (PROFILE_SHAPES and shape) else "{} [{}]".format(tag, shape)), flops, nbytes

    def __enter__(self):
        if self.on:
            self.i0 = _abi.lib().unetk_prof_mark()
        return self

    def __exit__(self, *exc):
        if self.on and exc[0] is None:
            PROFILE.append((self.tag, self.flops, self.i0, _abi.lib().unetk_prof_mark(), self.nbytes))
        return False


def profile_begin(reserve_launches=0):
    """bench.py: forget the kernel trace and pre-create its events (host work that must not sit in a timed region)."""
    check(_abi.lib().unetk_prof_reset(int(reserve_launches)), "prof_reset")


def profile_on(records):
    """Trace every launch of the library from here on and collect the op brackets into `records` (a list); None = off."""
    global PROFILE
    PROFILE = records
    check(_abi.lib().unetk_prof_enable(1 if records is not None else 0), "prof_enable")


def profile_read():
    """After a synchronize: (durations in ms, kernel names) of every traced launch since profile_begin()."""
    lib = _abi.lib()
    n = lib.unetk_prof_mark()
    ms = (ctypes.c_float * max(n, 1))()
    if n:
        check(lib.unetk_prof_read(0, n, ms), "prof_read")
    names, buf = [], ctypes.create_string_buffer(512)
    for i in range(n):
        check(lib.unetk_prof_name(i, buf, 512), "prof_name")
        names.append(buf.value.decode())
    return list(ms)[:n], names


class _NoTimer(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_TIMER = _NoTimer()


def _timed(tag_fn, flops, fmt=None, args=(), nbytes_fn=None):
    """Timer of a matrix kernel; tag_fn / the shape string / nbytes_fn are evaluated only when bench.py asked for events."""
    if PROFILE is None:
        return _NO_TIMER
    tag = tag_fn() if callable(tag_fn) else tag_fn
    return _Timed(tag, flops, fmt.format(*args) if fmt else None, nbytes_fn(tag) if nbytes_fn else 0)


def _timed_hbm(tag, t, passes, extra=0):
    """Timer of an HBM-bound pass moving `passes` x the bytes of tensor `t` (+ extra); a shared no-op unless bench.py asked
    for these (the byte count is not even computed then: these wrappers sit on the host path of every step)."""
    if PROFILE is None:
        return _NO_TIMER
    return _Timed(tag, 0.0, None, t.numel() * t.element_size() * passes + extra)


def _igemm_tag(cin, cout, bf16=False, h=0, n=1 << 20, w=1 << 10, nbr=False):
    """Kernel name of a conv3x3 forward / input-gradient launch (mirrors the dispatch of conv_igemm*.hip; bench labels)."""
    def cdiv(a, b):
        return -(-a // b)

    def big(bn):        # big_grid (conv_igemm.hip): 16 x 16 pixel tiles when they still give two blocks per CU
        return h % 16 == 0 and not (nbr and cin < 256) and n * (h // 16) * cdiv(w, 16) * (cout // bn) >= 512
    if bf16 and cin % 32 == 0 and cout % 32 == 0:
        bs = ",true>" if int(bf16) == _abi.BF16S else ">"          # <..., BS = true>: bf16 storage
        if int(bf16) == _abi.BF16S and h >= 24 and (cout % 128 == 0 or cout == 64):      # unetk_conv_bf16s_v3_ok (conv_igemm_bf16s.hip)
            bn = 128 if cout % 128 == 0 else 64
            if n * cdiv(h, 32) * cdiv(w, 16) * (cout // bn) >= 200:
                return "conv3x3_bf16s_kernel<{}{}>".format(bn // 16, ",nbr" if nbr else "")
        if cout % 128 == 0:
            tall = h >= 24 and n * cdiv(h, 32) * cdiv(w, 16) * (cout // 128) >= 200     # pick_bf16 (conv_igemm_bf16.hip)
            return ("conv3x3_igemm_bf16_kernel<4,2,4,2" if tall else "conv3x3_igemm_bf16_kernel<2,2,2,2") + bs
        if cout % 64 == 0:
            return ("conv3x3_igemm_bf16_kernel<4,1,2,2" if h >= 12 else "conv3x3_igemm_bf16_kernel<4,1,1,2") + bs
        return "conv3x3_igemm_bf16_kernel<4,1,2,1>"
    if cin % 16 == 0 and cout % 128 == 0:
        if n * cdiv(h, 8) * cdiv(w, 16) * (cout // 128) < 384:      # under-filled grid: half-height tiles
            return "conv3x3_igemm_kernel<2,2,1,2>"
        return "conv3x3_igemm_kernel<2,2,4,2>" if big(128) else "conv3x3_igemm_kernel<2,2,2,2>"
    if cin % 16 == 0 and cout % 64 == 0:
        return "conv3x3_igemm_kernel<4,1,2,2>" if big(64) else "conv3x3_igemm_kernel<4,1,1,2>"
    if cin % 16 == 0 and cout % 32 == 0:
        return "conv3x3_igemm_kernel<4,1,2,1>"
    if cout == 64 and 1 <= cin <= 5:            # first layers on the matrix pipe (conv_igemm.hip)
        return "conv3x3_c3_mfma_kernel"
    return "conv3x3_direct_kernel"


def alias(t, offset_elems=0, size=None, stride=None):
    """A fresh tensor object over t's storage (own autograd version counter).  The kernels write
    disjoint channel slices of shared concat buffers; autograd must not see those as in-place ops."""
    size = tuple(t.shape) if size is None else tuple(size)
    stride = tuple(t.stride()) if stride is None else tuple(stride)
    out = torch.empty(0, dtype=t.dtype, device=t.device)
    out.set_(t.untyped_storage(), t.storage_offset() + offset_elems, size, stride)
    return out


# ----------------------------------------------------------------------------- filter gradients on a second HIP stream
class _Side(object):
    """The filter gradient of a conv unit hangs off backward's critical chain (norm backward -> input gradient -> the
    previous unit's norm backward ...): nothing downstream reads dW before the optimiser.  With UNETK_SIDE_WGRAD=1 it is
    launched on a second HIP stream, ordered behind the main stream's dy, so the HBM-bound passes of the chain (norm
    backward, pool backward) can share the chip with it.  The main stream waits for the side stream at the end of the
    backward pass (autograd engine callback) and before a data-parallel bucket is all-reduced; dy and x are handed to the
    caching allocator as in use on the side stream (record_stream); the side stream has its own scratch buffer."""
    # Round 5: with the filter gradient queued BEFORE the unit's input gradient (SIDE_WGRAD_FIRST: the side stream then waits for
    # dy only and the two kernels really share the chip) the fp32 headline step gains 1.6 % in a same-call A/B (75.40 / 75.24 ->
    # 74.16 ms: the ~1.3 ms launches' tails fill each other), GUNet bs 8 nothing (21.44 vs 21.33-21.46), bf16 storage loses
    # (13.33 / 13.28 -> 13.42: its kernels hold all of a CU's LDS).  Default: on for fp32 units, off for the bf16 modes;
    # UNETK_SIDE_WGRAD=0 / 1 forces it off / on everywhere.
    mode = os.environ.get("UNETK_SIDE_WGRAD", "auto")
    enabled = mode != "0"
    paused = False          # bench.py: traced steps run on one stream (per-kernel durations that add up to the step)
    stream = None
    pending = False
    ws = _Workspace()


def side_wgrad_on(prec):
    """Does a 2-D conv unit of precision `prec` run its filter gradient on the side stream?"""
    if not _Side.enabled or _Side.paused:
        return False
    return _Side.mode == "1" or precision_of(prec) == _abi.FP32


def side_streams_pause(paused):
    """bench.py: pause (True) / resume (False) every use of the side stream (2-D and 3-D filter gradients)."""
    _Side.paused = bool(paused)


def side_join():
    """Order the current stream behind everything queued on the side stream (no-op when nothing is pending)."""
    if _Side.pending:
        torch.cuda.current_stream().wait_stream(_Side.stream)
        _Side.pending = False


def _wgrad_on_side(x, dy, bf16, dilation, out):
    if _Side.stream is None:
        _Side.stream = torch.cuda.Stream(device=x.device)
    side = _Side.stream
    side.wait_stream(torch.cuda.current_stream())          # dy has been queued on the main stream
    with torch.cuda.stream(side):
        dw = conv3x3_wgrad(x, dy, bf16=bf16, dilation=dilation, out=out, ws_pool=_Side.ws)
    dy.record_stream(side)
    x.record_stream(side)
    if not _Side.pending:
        _Side.pending = True
        torch.autograd.Variable._execution_engine.queue_callback(side_join)
    return dw


# 3-D layers: the filter gradient runs on the side stream BESIDE the input gradient.  UNet3D at one or two patches per GPU is a
# sequence of 100-800 us launches of a few hundred tiles whose tails leave CUs idle (and the bridge's 27-54 blocks never fill the
# chip); two such launches side by side share it.  Same kernels, same results bit for bit.  Measured in one call, 96^3: 20.52 ->
# 20.29 ms at one patch, 38.0 -> 37.6 at two (the bridge alone: no gain; 10 x 256 x 256: no change).  Threshold in output
# voxels of the layer (0 = never); UNETK_SIDE_WGRAD3D overrides.
# Round 5 (the advisor's finding): round 4 queued the side-stream launch AFTER the input gradient, so the side stream's
# wait_stream(main) made the filter gradient wait for that input gradient too -- it ran beside the NEXT unit's norm backward, not
# beside "its" input gradient as the text above says.  *_FIRST (default): it is queued BEFORE the input gradient, behind dy only.
# Same-call A/Bs: UNet3D 96^3 at one patch 20.51 -> 19.97 ms, at two 37.83 -> 36.93; the fp32 2-D headline 75.3 -> 74.2 ms.
SIDE_WGRAD3D_VOXELS = int(os.environ.get("UNETK_SIDE_WGRAD3D", str(1 << 20)))
SIDE_WGRAD_FIRST = os.environ.get("UNETK_SIDE_WGRAD_FIRST", "1") == "1"          # the 2-D units (see _Side)
SIDE_WGRAD3D_FIRST = os.environ.get("UNETK_SIDE_WGRAD3D_FIRST", "1") == "1"      # the 3-D units (Conv3dNormRelu.backward)


def _wgrad3d_on_side(x, dy, d, out):
    if _Side.stream is None:
        _Side.stream = torch.cuda.Stream(device=x.device)
    side = _Side.stream
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dw = conv3d_wgrad(x, dy, d, out=out, ws_pool=_Side.ws)
    dy.record_stream(side)
    x.record_stream(side)
    if not _Side.pending:
        _Side.pending = True
        torch.autograd.Variable._execution_engine.queue_callback(side_join)
    return dw


# ----------------------------------------------------------------------------- in-place parameter gradients
class _GradSink(object):
    """Parameter gradients written by the backward kernels straight into the variable's slice of the flat gradient
    buffer (ParamStore.grad) instead of a fresh tensor that autograd then adds to it: 65 tiny add launches per step
    gone.  A slot is written in place at most once between two new_step() calls (ParamStore.zero_grad), and only when
    the variable has exactly ONE recorded use (`uses`, counted by the forwards that end up on an autograd tape; the count
    restarts with the first recorded forward after a backward -- Solver.__call__ runs zero_grad BETWEEN the forward and
    its backward, so zero_grad cannot be the reset point): a variable shared by two ops takes the returned-tensor path
    for every use, i.e. autograd sums the contributions and the post-accumulate hook fires once, after the sum -- the
    data-parallel buckets must never see an arrival that precedes the variable's last contribution.  A denied slot is
    always safe (autograd accumulates).  `hooks` maps a slot to the callback the buckets would have got from a
    post-accumulate hook."""
    written = set()
    uses = {}
    tape = True         # is the forward being run recorded on an autograd tape (set by _Op.apply)
    stale = False       # a backward has run since `uses` was last cleared: the next recorded forward starts a fresh count
    hooks = {}
    enabled = True


class _Op(torch.autograd.Function):
    """Base of the libunetk autograd nodes: notes, before the forward runs (inside it grad mode is always off and
    needs_input_grad is set regardless), whether a tape is recording -- an evaluation forward under torch.no_grad() must
    not count as a use of its parameters (grad_sink)."""

    @classmethod
    def apply(cls, *args, **kwargs):
        # saved and restored: an op applied INSIDE another op's forward (the SE gate's FullyConnected under
        # torch.enable_grad()) must not leave its own answer behind for the outer forward's grad_sink() calls
        prev = _GradSink.tape
        _GradSink.tape = torch.is_grad_enabled()
        try:
            return super(_Op, cls).apply(*args, **kwargs)
        finally:
            _GradSink.tape = prev


def new_step(bufs=None):
    """Start of a step for the gradient buffers `bufs` (a ParamStore's flat gradient tensors; None = every slot): their
    slots may be written in place again.  Another model's slots keep their state (its own zero_grad resets them)."""
    if bufs is None:
        _GradSink.written.clear()
    else:
        spans = [(b.data_ptr(), b.data_ptr() + b.numel() * b.element_size()) for b in bufs]
        _GradSink.written.difference_update([k for k in _GradSink.written if any(lo <= k < hi for lo, hi in spans)])
    FUSED_NBR.clear()
    if _Side.pending:           # a backward that raised left filter gradients on the side stream un-joined
        side_join()


def grad_sink(p, ctx=None):
    """The flat-gradient slot of leaf parameter `p` when a backward kernel may write it in place, else None.  Called from
    an op's forward with its autograd context: a forward no tape records (torch.no_grad(): evaluation) is not a use."""
    if not _GradSink.enabled or p is None or not p.is_leaf or not p.requires_grad:
        return None
    if ctx is not None and not _GradSink.tape:
        return None
    g = p.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.shape != p.shape:
        return None
    if _GradSink.stale:
        _GradSink.uses.clear()
        _GradSink.stale = False
    k = g.data_ptr()
    _GradSink.uses[k] = _GradSink.uses.get(k, 0) + 1
    return g


def _take(slot):
    """Claim `slot` for an in-place write in this step (None if absent, already written, or used more than once)."""
    _GradSink.stale = True
    if slot is None:
        return None
    k = slot.data_ptr()
    if k in _GradSink.written or _GradSink.uses.get(k, 0) != 1:
        return None
    _GradSink.written.add(k)
    return slot


def _ret(value, slot):
    """What a backward returns for a parameter: None when its gradient already sits in `slot` (the bucket hook fires
    here, after the kernel was queued), else the tensor for autograd to accumulate."""
    if slot is None:
        return value
    hook = _GradSink.hooks.get(slot.data_ptr())
    if hook is not None:
        hook()
    return None


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _abi.UnetkError("libunetk ops need device tensors (got a CPU tensor); "
                                  "there is no CPU path in the product")


def _pix_stride(t):
    """Pixel stride (floats) of an NHWC tensor whose channel dim is contiguous."""
    assert t.dim() == 4 and t.stride(3) == 1, "NHWC with contiguous channels expected"
    s = t.stride(2)
    assert t.stride(1) == t.shape[2] * s and t.stride(0) == t.shape[1] * t.shape[2] * s, \
        "only a channel-slice view of a dense NHWC buffer is supported"
    return s


# ----------------------------------------------------------------------------- raw op wrappers
def conv_uses_bf16(cin, cout):
    """UNETK_BF16 kernels exist for Cin % 32 == 0 and Cout % 32 == 0; other layers (the first conv) stay fp32."""
    return cin % 32 == 0 and cout % 32 == 0


def precision_of(flag):
    """NormSpec.bf16 / --compute_dtype -> UNETK_FP32 (0) | UNETK_BF16 (1: bf16 MFMA operands, fp32 tensors) |
    UNETK_BF16S (2: bf16 MFMA + bf16 STORAGE of activations and activation gradients, include/unetk.h)."""
    if flag is True:
        return _abi.BF16
    return int(flag or 0)


def storage_dtype(prec):
    return torch.bfloat16 if int(prec) == _abi.BF16S else torch.float32


def _storage_of(t):
    """Descriptor `storage` value of an activation tensor."""
    if t.dtype == torch.bfloat16:
        return _abi.BF16S
    assert t.dtype == torch.float32, t.dtype
    return _abi.FP32


def conv3x3_pack(w, want_dgrad=True, bf16=False):
    _require_cuda(w)
    kh, kw, cin, cout = w.shape
    assert kh == 3 and kw == 3
    prec = precision_of(bf16)

    def build():
        if prec:
            wp_f = torch.empty(9 * cin * cout, dtype=torch.bfloat16, device=w.device)
            wp_d = torch.empty_like(wp_f) if want_dgrad else None
            if prec == _abi.BF16S:          # output channels pair-permuted inside 64-blocks (conv_igemm_bf16.hip)
                check(_abi.lib().unetk_conv3x3_pack_bf16s(ptr(w), cin, cout, ptr(wp_f), ptr(wp_d), stream_ptr()),
                      "conv3x3_pack_bf16s")
            else:
                check(_abi.lib().unetk_conv3x3_pack_bf16(ptr(w), cin, cout, ptr(wp_f), ptr(wp_d), stream_ptr()),
                      "conv3x3_pack_bf16")
            kind, perm = PK_CONV_BF16, 1 if prec == _abi.BF16S else 0
        else:
            wp_f = torch.empty(9 * cin * cout, dtype=torch.float32, device=w.device)
            wp_d = torch.empty_like(wp_f) if want_dgrad else None
            check(_abi.lib().unetk_conv3x3_pack(ptr(w), cin, cout, ptr(wp_f), ptr(wp_d), stream_ptr()), "conv3x3_pack")
            kind, perm = PK_CONV_F32, 0
        return wp_f, wp_d, [(kind, perm, cin, cout, w.data_ptr(), wp_f.data_ptr(), wp_d.data_ptr() if want_dgrad else None)]
    return PACKS.get(w, ("c3", prec, bool(want_dgrad)), build)


def conv_uses_mfma(cin, cout):
    return cin % 16 == 0 and cout % 32 == 0


def conv3x3_fwd(x, w, cout, want_stats=True, y=None, bf16=False, dilation=1):
    """x NHWC (pixel-strided ok); w = packed filter if conv_uses_mfma(cin, cout) else raw HWIO.
    bf16 = precision (precision_of): under UNETK_BF16S x is bf16 (fp32 for the direct first-layer kernel) and y is bf16."""
    _require_cuda(x, w)
    n, h, wd, cin = x.shape
    prec = precision_of(bf16)
    if y is None:
        y = torch.empty((n, h, wd, cout), dtype=storage_dtype(prec), device=x.device)
    if prec == _abi.BF16S:
        assert y.dtype == torch.bfloat16 and x.dtype == (torch.bfloat16 if conv_uses_mfma(cin, cout) else torch.float32), \
            (x.dtype, y.dtype, cin, cout)
    else:
        assert x.dtype == torch.float32 and y.dtype == torch.float32
    d = ConvDesc(n, h, wd, cin, cout, _pix_stride(x), _pix_stride(y), prec, int(dilation))
    stats = None
    rows = 0
    if want_stats:
        rows = _abi.lib().unetk_conv3x3_stat_rows(ctypes.byref(d))
        if rows <= 0:
            check(rows, "conv3x3_stat_rows")
        stats = torch.empty((2, rows, cout), dtype=torch.float32, device=x.device)
    nws = _abi.lib().unetk_conv3x3_ws_bytes(ctypes.byref(d)) if prec == _abi.FP32 else 0     # stream-K scratch (small planes)
    ws = WORKSPACE.get(nws, x.device) if nws else None
    # the first layer's direct kernel is HBM-bound (writes 64 channels per pixel from 3): reported by bytes as well
    with _timed(lambda: _igemm_tag(cin, cout, bf16, h, n, wd), 18.0 * n * h * wd * cin * cout, "fwd {}x{}x{} {}->{}", (n, h, wd, cin, cout),
                lambda tag: (x.numel() * x.element_size() + y.numel() * y.element_size()) if tag in ("conv3x3_direct_kernel", "conv3x3_c3_mfma_kernel") else 0):
        check(_abi.lib().unetk_conv3x3_fwd_ws(ctypes.byref(d), ptr(x), ptr(w), ptr(y), ptr(stats), ptr(ws), nws,
                                              stream_ptr()), "conv3x3_fwd")
    return y, stats, rows


def conv3x3_fwd_affine(x, w, cout, scale, shift, z=None, pool=False, bf16=False):
    """Inference: z = relu(conv3x3(x, w) * scale + shift) [and max_pool2d(z, 2, 2)] in one kernel (unetk_conv3x3_fwd_affine);
    w as for conv3x3_fwd.  Returns (z, pooled or None); raises UNETK_E_UNSUPPORTED shapes (ask conv3x3_fwd_affine_ok first)."""
    _require_cuda(x, w)
    n, h, wd, cin = x.shape
    prec = precision_of(bf16)
    if z is None:
        z = torch.empty((n, h, wd, cout), dtype=storage_dtype(prec), device=x.device)
    d = ConvDesc(n, h, wd, cin, cout, _pix_stride(x), _pix_stride(z), prec, 1)
    pooled = torch.empty((n, h // 2, wd // 2, cout), dtype=z.dtype, device=x.device) if pool else None
    nws = _abi.lib().unetk_conv3x3_ws_bytes(ctypes.byref(d))
    ws = WORKSPACE.get(nws, x.device) if nws else None
    check(_abi.lib().unetk_conv3x3_fwd_affine(ctypes.byref(d), ptr(x), ptr(w), ptr(scale), ptr(shift), ptr(z), ptr(pooled), cout,
                                              ptr(ws), nws, stream_ptr()), "conv3x3_fwd_affine")
    return z, pooled


def conv3x3_fwd_affine_ok(n, h, wd, cin, cout, pool=False, bf16=False):
    d = ConvDesc(n, h, wd, cin, cout, cin, cout, precision_of(bf16), 1)
    return _abi.lib().unetk_conv3x3_fwd_affine_ok(ctypes.byref(d), 1 if pool else 0) == 1


def conv3x3_dgrad(dy, wp_dgrad, cin, x_stride=None, dx=None, bf16=False, dilation=1, producer=None):
    """producer = (y, aff, per_sample) of the unit whose activation is this conv's input: the kernel then also emits that
    unit's norm-backward reduction (unetk_conv3x3_dgrad_nbr) and the partials are left in FUSED_NBR for its backward."""
    _require_cuda(dy, wp_dgrad)
    n, h, wd, cout = dy.shape
    prec = precision_of(bf16)
    assert dy.dtype == storage_dtype(prec), (dy.dtype, prec)
    if dx is None:
        dx = torch.empty((n, h, wd, cin), dtype=dy.dtype, device=dy.device)
    d = ConvDesc(n, h, wd, cin, cout, _pix_stride(dx), _pix_stride(dy), prec, int(dilation))
    if producer is not None and FUSE_NBR:
        py, paff, per_sample = producer
        rows = _abi.lib().unetk_conv3x3_dgrad_nbr_rows(ctypes.byref(d))
        if rows > 0 and py.dtype == dx.dtype and tuple(py.shape) == tuple(dx.shape) and py.is_contiguous():
            part = torch.empty((2, rows, cin), dtype=torch.float32, device=dy.device)
            with _timed(lambda: _igemm_tag(cout, cin, bf16, h, n, wd, nbr=True), 18.0 * n * h * wd * cin * cout,
                        "dgrad+nbr {}x{}x{} {}->{}", (n, h, wd, cout, cin)):
                check(_abi.lib().unetk_conv3x3_dgrad_nbr(ctypes.byref(d), ptr(dy), ptr(wp_dgrad), ptr(dx), ptr(py), cin,
                                                         ptr(paff[2]), ptr(paff[3]), ptr(paff[0]), ptr(paff[1]),
                                                         1 if per_sample else 0, ptr(part), stream_ptr()),
                      "conv3x3_dgrad_nbr")
            # dx._version: when the producer's activation has a SECOND consumer, autograd sums the gradients IN PLACE into
            # this tensor (same data_ptr, same shape) -- the partials computed from this dx alone would be stale; any in-place
            # accumulation bumps the version counter, and the producer's backward then ignores the entry
            FUSED_NBR[dx.data_ptr()] = (py.data_ptr(), tuple(dx.shape), part, rows, dx._version)
            return dx
    nws = _abi.lib().unetk_conv3x3_ws_bytes(ctypes.byref(d)) if prec == _abi.FP32 else 0
    ws = WORKSPACE.get(nws, dy.device) if nws else None
    with _timed(lambda: _igemm_tag(cout, cin, bf16, h, n, wd), 18.0 * n * h * wd * cin * cout, "dgrad {}x{}x{} {}->{}", (n, h, wd, cout, cin)):
        check(_abi.lib().unetk_conv3x3_dgrad_ws(ctypes.byref(d), ptr(dy), ptr(wp_dgrad), ptr(dx), ptr(ws), nws, stream_ptr()),
              "conv3x3_dgrad")
    return dx


def conv3x3_wgrad(x, dy, bf16=False, dilation=1, out=None, ws_pool=None):
    _require_cuda(x, dy)
    n, h, wd, cin = x.shape
    cout = dy.shape[3]
    prec = precision_of(bf16)
    if prec == _abi.BF16S:
        assert dy.dtype == torch.bfloat16 and x.dtype == (torch.bfloat16 if cin % 64 == 0 else torch.float32), (x.dtype, dy.dtype)
    d = ConvDesc(n, h, wd, cin, cout, _pix_stride(x), _pix_stride(dy), prec, int(dilation))
    nbytes = _abi.lib().unetk_conv3x3_wgrad_ws_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise _abi.UnetkError("conv3x3_wgrad: unsupported shape Cin={} Cout={}".format(cin, cout))
    ws = (ws_pool or WORKSPACE).get(nbytes, x.device)
    dw = out if out is not None else torch.empty((3, 3, cin, cout), dtype=torch.float32, device=x.device)
    assert tuple(dw.shape) == (3, 3, cin, cout) and dw.is_contiguous()
    tag = "conv3x3_wgrad_kernel(+slab_reduce)" if cin % 64 == 0 else "conv3x3_wgrad_c3_kernel(+slab_reduce)"
    if bf16 and cin % 32 == 0:
        tag = "conv3x3_wgrad_kernel<bf16>(+slab_reduce)"
    if prec == _abi.BF16S:
        tag = "conv3x3_wgrad_bf16s_kernel(+slab_reduce)" if cin % 64 == 0 else "conv3x3_wgrad_c3_bf16s_kernel(+slab_reduce)"
    with _timed(tag, 18.0 * n * h * wd * cin * cout, "{}x{}x{} {}->{}", (n, h, wd, cin, cout)):
        check(_abi.lib().unetk_conv3x3_wgrad(ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(ws), nbytes,
                                             stream_ptr()), "conv3x3_wgrad")
    return dw


def conv3d_desc(x_shape, cout, kd, stride, x_stride=None, y_stride=None, live8=None):
    """live8 = (cin mask, cout mask) of a channel-padded filter (NetworksV2/padded.py live8_masks): bit i = channels
    [8 i, 8 i + 8) hold a real channel; the kernels skip the dead groups of their contraction axis (include/unetk.h)."""
    n, dd, h, w, cin = x_shape
    sd, sh, sw = stride
    assert sh == sw, "H and W strides must match"
    d = Conv3dDesc(n, dd, h, w, cin, cout, kd, sd, sh, x_stride or cin, y_stride or cout)
    if live8 is not None:
        for field, m in ((d.cin_live8, live8[0]), (d.cout_live8, live8[1])):
            field[0], field[1] = m & 0xFFFFFFFF, (m >> 32) & 0xFFFFFFFF
    return d


def conv3d_out_shape(d):
    do, ho, wo = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    check(_abi.lib().unetk_conv3d_out_dims(ctypes.byref(d), ctypes.byref(do), ctypes.byref(ho), ctypes.byref(wo)),
          "conv3d_out_dims")
    return (d.N, do.value, ho.value, wo.value, d.Cout)


def conv3d_pack(w, want_dgrad=True):
    """w: TF DHWIO [kd,3,3,Cin,Cout] (Cin % 4 == 0 for the MFMA path)."""
    _require_cuda(w)
    kd, kh, kw, cin, cout = w.shape
    assert kh == 3 and kw == 3

    def build():
        wp_f = torch.empty(kd * 9 * cin * cout, dtype=torch.float32, device=w.device)
        wp_d = torch.empty_like(wp_f) if want_dgrad else None
        check(_abi.lib().unetk_conv3d_pack(ptr(w), kd, cin, cout, ptr(wp_f), ptr(wp_d), stream_ptr()), "conv3d_pack")
        tap = 9 * cin * cout * 4            # bytes per depth tap, source and packs alike
        items = [(PK_CONV_F32, 0, cin, cout, w.data_ptr() + a * tap, wp_f.data_ptr() + a * tap,
                  (wp_d.data_ptr() + a * tap) if want_dgrad else None) for a in range(kd)]
        return wp_f, wp_d, items
    return PACKS.get(w, ("c3d", 0, bool(want_dgrad)), build)


def _ws3d(d, device):
    nbytes = _abi.lib().unetk_conv3d_ws_bytes(ctypes.byref(d))
    return WORKSPACE.get(nbytes, device), nbytes


def conv3d_fwd(x, w, d, want_stats=True):
    """x [N,D,H,W,Cin] dense; w = packed filter when conv_uses_mfma(Cin, Cout) else raw DHWIO (kd == 1)."""
    _require_cuda(x, w)
    assert x.stride(-1) == 1 and x.stride(-2) == d.x_stride        # dense, or a channel slice of a concat buffer
    y = torch.empty(conv3d_out_shape(d), dtype=torch.float32, device=x.device)
    stats, rows = None, 0
    if want_stats:
        rows = _abi.lib().unetk_conv3d_stat_rows(ctypes.byref(d))
        if rows <= 0:
            check(rows, "conv3d_stat_rows")
        stats = torch.empty((2, rows, d.Cout), dtype=torch.float32, device=x.device)
    ws, nbytes = _ws3d(d, x.device)
    flops = 2.0 * y.numel() / d.Cout * d.kd * 9 * d.Cin * d.Cout
    with _Timed("conv3d_fwd", flops, "k{}s{}{} {}".format(d.kd, d.sd, d.shw, tuple(x.shape))):
        check(_abi.lib().unetk_conv3d_fwd(ctypes.byref(d), ptr(x), ptr(w), ptr(y), ptr(stats), ptr(ws), nbytes,
                                          stream_ptr()), "conv3d_fwd")
    return y, stats, rows


def conv3d_dgrad(dy, wp_dgrad, d):
    assert dy.is_contiguous()
    dx = torch.empty((d.N, d.D, d.H, d.W, d.Cin), dtype=torch.float32, device=dy.device)
    ws, nbytes = _ws3d(d, dy.device)
    flops = 2.0 * dy.numel() / d.Cout * d.kd * 9 * d.Cin * d.Cout
    with _Timed("conv3d_dgrad", flops, "k{}s{}{} {}".format(d.kd, d.sd, d.shw, tuple(dx.shape))):
        check(_abi.lib().unetk_conv3d_dgrad(ctypes.byref(d), ptr(dy), ptr(wp_dgrad), ptr(dx), ptr(ws), nbytes,
                                            stream_ptr()), "conv3d_dgrad")
    return dx


def conv3d_wgrad(x, dy, d, out=None, ws_pool=None):
    assert x.stride(-2) == d.x_stride and dy.is_contiguous()
    dw = out if out is not None else torch.empty((d.kd, 3, 3, d.Cin, d.Cout), dtype=torch.float32, device=x.device)
    assert tuple(dw.shape) == (d.kd, 3, 3, d.Cin, d.Cout) and dw.is_contiguous()
    nbytes = _abi.lib().unetk_conv3d_ws_bytes(ctypes.byref(d))
    ws = (ws_pool or WORKSPACE).get(nbytes, x.device)
    flops = 2.0 * dy.numel() / d.Cout * d.kd * 9 * d.Cin * d.Cout
    with _Timed("conv3d_wgrad", flops, "k{}s{}{} {}".format(d.kd, d.sd, d.shw, tuple(x.shape))):
        check(_abi.lib().unetk_conv3d_wgrad(ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(ws), nbytes, stream_ptr()),
              "conv3d_wgrad")
    return dw


def norm_desc(y_shape, per_sample, z_stride=None, guide_ch=0, gw_stride=0, gw_coff=0):
    n, c = y_shape[0], y_shape[-1]
    hw = 1
    for s in y_shape[1:-1]:
        hw *= s
    return NormDesc(n, hw, c, 1 if per_sample else 0, z_stride if z_stride is not None else c, guide_ch,
                    gw_stride, gw_coff)


def norm_finalize(d, stats, rows, gamma, beta, eps, decay, training, moving_mean, moving_var, device):
    groups = d.N if d.per_sample else 1
    out = torch.empty((4, groups, d.C), dtype=torch.float32, device=device)   # mean, rstd, scale, shift
    nbytes = _abi.lib().unetk_norm_finalize_ws_bytes(ctypes.byref(d), max(rows, groups))
    ws = WORKSPACE.get(nbytes, device)
    check(_abi.lib().unetk_norm_finalize(ctypes.byref(d), ptr(stats), rows, ptr(gamma), ptr(beta), eps, decay,
                                         1 if training else 0, ptr(moving_mean), ptr(moving_var), ptr(out[0]),
                                         ptr(out[1]), ptr(out[2]), ptr(out[3]), ptr(ws), nbytes, stream_ptr()),
          "norm_finalize")
    return out


def norm_apply_relu(d, y, aff, z, guide=None, gw=None, gb=None, den=None):
    assert y.is_contiguous() and y.dtype == z.dtype
    d.storage = _storage_of(y)
    if den is not None:
        assert den.is_contiguous() and tuple(den.shape) == (d.N, d.C), (tuple(den.shape), d.N, d.C)
    with _timed_hbm("norm_apply_relu", y, 2):          # read y, write z
        check(_abi.lib().unetk_norm_apply_relu(ctypes.byref(d), ptr(y), ptr(aff[2]), ptr(aff[3]), ptr(den), ptr(guide),
                                               ptr(gw), ptr(gb), ptr(z), stream_ptr()), "norm_apply_relu")
    return z


def norm_relu_bwd(d, y, dz, aff, has_gamma, has_beta, guide=None, gw=None, gb=None, den=None, out_gamma=None,
                  out_beta=None, pre=None):
    dev = y.device
    assert dz.dtype == y.dtype, (dz.dtype, y.dtype)
    d.storage = _storage_of(y)
    dy = torch.empty_like(y)
    dden = torch.empty_like(den) if den is not None else None
    dgamma = (out_gamma if out_gamma is not None else torch.empty((d.C,), dtype=torch.float32, device=dev)) if has_gamma else None
    dbeta = (out_beta if out_beta is not None else torch.empty((d.C,), dtype=torch.float32, device=dev)) if has_beta else None
    blk = (4,) if d.guide_leaky == 3 else ()       # guide_leaky == 3: gb is the block [bias, slope+, slope-, post-shift] and so is its gradient
    if d.guide_per_sample:
        dgw = torch.empty((d.N, d.guide_ch, d.C), dtype=torch.float32, device=dev) if d.guide_ch else None
        dgb = torch.empty((d.N,) + blk + (d.C,), dtype=torch.float32, device=dev) if (d.guide_ch or gb is not None) else None
    else:
        dgw = torch.empty((d.guide_ch, d.C), dtype=torch.float32, device=dev) if d.guide_ch else None
        dgb = torch.empty(blk + (d.C,), dtype=torch.float32, device=dev) if (d.guide_ch or gb is not None) else None
    nbytes = _abi.lib().unetk_norm_bwd_ws_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise _abi.UnetkError("norm_relu_bwd: unsupported channel count {}".format(d.C))
    ws = WORKSPACE.get(nbytes, dev)
    pre_part, pre_rows = pre if pre is not None else (None, 0)
    # reduction pass reads (dz, y) unless its partials came from the producing kernel; the apply pass reads (dz, y), writes dy
    with _timed_hbm("norm_relu_bwd(apply only)" if pre is not None else "norm_relu_bwd(reduce+apply)", y, 3 if pre is not None else 5):
        check(_abi.lib().unetk_norm_relu_bwd_pre(ctypes.byref(d), ptr(y), ptr(dz), _pix_stride_nd(dz), ptr(aff[2]), ptr(aff[3]),
                                                 ptr(aff[0]), ptr(aff[1]), ptr(den), ptr(guide), ptr(gw), ptr(gb), ptr(dy),
                                                 ptr(dgamma), ptr(dbeta), ptr(dden), ptr(dgw), ptr(dgb), ptr(pre_part),
                                                 int(pre_rows), ptr(ws), nbytes, stream_ptr()),
              "norm_relu_bwd")
    if den is not None:
        return dy, dgamma, dbeta, dgw, dgb, dden
    return dy, dgamma, dbeta, dgw, dgb


def norm_apply_relu_pool(d, y, aff, z):
    """z = relu(y * scale + shift) and max_pool2d(z, 2, 2) in one pass (unetk_norm_apply_relu_pool); returns the pooled tensor."""
    d.storage = _storage_of(y)
    n, h, w, c = y.shape
    pooled = torch.empty((n, h // 2, w // 2, c), dtype=y.dtype, device=y.device)
    with _timed_hbm("norm_apply_relu_pool", y, 2.25):
        check(_abi.lib().unetk_norm_apply_relu_pool(ctypes.byref(d), int(w), ptr(y), ptr(aff[2]), ptr(aff[3]), ptr(z), ptr(pooled),
                                                    stream_ptr()), "norm_apply_relu_pool")
    return pooled


def norm_relu_bwd_pool(d, y, dskip, dp, aff, has_gamma, has_beta, out_gamma=None, out_beta=None):
    """norm_relu_bwd of a plain unit whose activation feeds max_pool2d AND the skip connection, with the pool's backward
    folded in (unetk_norm_relu_bwd_pool): dz = dskip + route(dp) is never written.  y [N, H, W, C]."""
    dev = y.device
    assert dskip.dtype == y.dtype and dp.dtype == y.dtype and dp.is_contiguous()
    d.storage = _storage_of(y)
    dy = torch.empty_like(y)
    dgamma = (out_gamma if out_gamma is not None else torch.empty((d.C,), dtype=torch.float32, device=dev)) if has_gamma else None
    dbeta = (out_beta if out_beta is not None else torch.empty((d.C,), dtype=torch.float32, device=dev)) if has_beta else None
    nbytes = _abi.lib().unetk_norm_bwd_ws_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise _abi.UnetkError("norm_relu_bwd_pool: unsupported channel count {}".format(d.C))
    ws = WORKSPACE.get(nbytes, dev)
    # both passes read (y, dskip) and a quarter-size dp; the apply pass writes dy
    with _timed_hbm("norm_relu_bwd_pool(reduce+apply)", y, 5.5):
        check(_abi.lib().unetk_norm_relu_bwd_pool(ctypes.byref(d), int(y.shape[2]), ptr(y), ptr(dskip), _pix_stride_nd(dskip),
                                                  ptr(dp), ptr(aff[2]), ptr(aff[3]), ptr(aff[0]), ptr(aff[1]), ptr(dy),
                                                  ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, stream_ptr()),
              "norm_relu_bwd_pool")
    return dy, dgamma, dbeta


norm_relu_bwd_nd = norm_relu_bwd     # rank-agnostic (dz pixel stride = stride of the second-to-last axis)


def sobel_concat(x, ch):
    """InterUNet --img_grad: concat(x, sobel_dy(x[..., ch]), sobel_dx(x[..., ch])) (InterUNet.py:105-109); no gradient."""
    _require_cuda(x)
    x = x.to(torch.float32).contiguous()
    n, h, w, c = x.shape
    out = torch.empty((n, h, w, c + 2), dtype=torch.float32, device=x.device)
    check(_abi.lib().unetk_sobel_concat(ptr(x), ptr(out), n, h, w, c, int(ch), stream_ptr()), "sobel_concat")
    return out


def image_gradients(x):
    """--img_grad: concat(x, dy, dx) along channels (UNet.py:69-71); the input needs no gradient."""
    _require_cuda(x)
    n, h, w, c = x.shape
    out = torch.empty((n, h, w, 3 * c), dtype=torch.float32, device=x.device)
    check(_abi.lib().unetk_image_gradients(ptr(x.contiguous()), ptr(out), n, h, w, c, stream_ptr()), "image_gradients")
    return out


def flip_axpy(x, out=None, flip_h=False, flip_w=False, scale=1.0, accumulate=False):
    """out (+)= scale * flip(x) over H and/or W of an NHWC tensor (mirror TTA, evaluator_liver.py:648-655)."""
    _require_cuda(x)
    x = x.contiguous()
    n, h, w, c = x.shape
    if out is None:
        out = torch.empty_like(x)
        accumulate = False
    assert out.is_contiguous() and out.shape == x.shape and out.data_ptr() != x.data_ptr()
    check(_abi.lib().unetk_flip_axpy(ptr(x), ptr(out), n, h, w, c, int(bool(flip_h)), int(bool(flip_w)), float(scale),
                                     int(bool(accumulate)), stream_ptr()), "flip_axpy")
    return out


def avgpool2_fwd(x):
    _require_cuda(x)
    n, h, w, c = x.shape
    p = torch.empty((n, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
    check(_abi.lib().unetk_avgpool2_fwd(ptr(x.contiguous()), ptr(p), n, h, w, c, stream_ptr()), "avgpool2_fwd")
    return p


def maxpool2_fwd(x):
    _require_cuda(x)
    n, h, w, c = x.shape
    p = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    fn = _abi.lib().unetk_maxpool2_fwd_bf16 if _storage_of(x) == _abi.BF16S else _abi.lib().unetk_maxpool2_fwd
    with _timed_hbm("maxpool2_fwd", p, 5):            # reads x (4 p), writes p
        check(fn(ptr(x), _pix_stride(x), ptr(p), n, h, w, c, stream_ptr()), "maxpool2_fwd")
    return p


def maxpool2_bwd(x, p, dp, add=None):
    """dx = route(dp) [+ add]; `add` may be a channel slice of a wider NHWC buffer (the concat buffer's gradient)."""
    n, h, w, c = x.shape
    dx = torch.empty((n, h, w, c), dtype=x.dtype, device=x.device)
    assert dp.dtype == x.dtype and p.dtype == x.dtype
    if add is not None:
        assert tuple(add.shape) == (n, h, w, c) and add.stride(3) == 1 and add.dtype == x.dtype
    fn = _abi.lib().unetk_maxpool2_bwd_bf16 if _storage_of(x) == _abi.BF16S else _abi.lib().unetk_maxpool2_bwd
    # reads x, p, dp (+ the skip gradient), writes dx
    with _timed_hbm("maxpool2_bwd", p, (4 * 3 if add is not None else 4 * 2) + 2):
        check(fn(ptr(x), _pix_stride(x), ptr(p), ptr(dp.contiguous()), ptr(add),
                 _pix_stride(add) if add is not None else 0, ptr(dx), n, h, w, c, stream_ptr()), "maxpool2_bwd")
    return dx


def deconv2x2_pack(w, bf16=False):
    _require_cuda(w)
    kh, kw, cout, cin = w.shape
    assert kh == 2 and kw == 2
    prec = precision_of(bf16)

    def build():
        if prec:
            wp_f = torch.empty(4 * cin * cout, dtype=torch.bfloat16, device=w.device)
            wp_d = torch.empty_like(wp_f)
            if prec == _abi.BF16S:
                check(_abi.lib().unetk_deconv2x2_pack_bf16s(ptr(w), cin, cout, ptr(wp_f), ptr(wp_d), stream_ptr()),
                      "deconv2x2_pack_bf16s")
            else:
                check(_abi.lib().unetk_deconv2x2_pack_bf16(ptr(w), cin, cout, ptr(wp_f), ptr(wp_d), stream_ptr()),
                      "deconv2x2_pack_bf16")
            kind, perm = PK_DECONV_BF16, 1 if prec == _abi.BF16S else 0
        else:
            wp_f = torch.empty(4 * cin * cout, dtype=torch.float32, device=w.device)
            wp_d = torch.empty_like(wp_f)
            check(_abi.lib().unetk_deconv2x2_pack(ptr(w), cin, cout, ptr(wp_f), ptr(wp_d), stream_ptr()), "deconv2x2_pack")
            kind, perm = PK_DECONV_F32, 0
        return wp_f, wp_d, [(kind, perm, cin, cout, w.data_ptr(), wp_f.data_ptr(), wp_d.data_ptr())]
    return PACKS.get(w, ("d2", prec, True), build)


def deconv2x2_fwd(x, wp_fwd, bias, cat, coff, cout, bf16=False):
    n, h, w, cin = x.shape
    assert x.is_contiguous()
    prec = precision_of(bf16)
    assert x.dtype == storage_dtype(prec) and cat.dtype == x.dtype, (x.dtype, cat.dtype, prec)
    d = DeconvDesc(n, h, w, cin, cout, _pix_stride(cat), coff, prec)
    tag = {0: "pw_gemm_kernel<fwd>", 1: "pw_gemm_bf16_kernel<fwd>", 2: "pw_gemm_bf16_kernel<fwd,bs>"}[prec]
    with _timed(tag, 8.0 * n * h * w * cin * cout, "{}x{}x{} {}->{}", (n, h, w, cin, cout)):
        check(_abi.lib().unetk_deconv2x2_fwd(ctypes.byref(d), ptr(x), ptr(wp_fwd), ptr(bias), ptr(cat), stream_ptr()),
              "deconv2x2_fwd")
    return cat


# Round 5: the transposed conv's filter gradient on the side stream as well.  Its backward is ReLU mask -> input gradient -> filter
# gradient in one C call; only dx is on backward's critical chain.  unetk_deconv*_bwd_parts does the first two on the main stream
# and the filter gradient (which reads the masked gradient the first part left in the scratch buffer) on the side stream, beside
# the next unit's norm backward / input gradient.  The scratch buffer is then the op's own (the shared WORKSPACE is rewritten by
# the next op on the main stream).  Needs the in-place gradient slot (otherwise autograd would add dw on the main stream at once).
SIDE_DECONV = os.environ.get("UNETK_SIDE_DECONV", "1") == "1"


def _deconv_bwd_call(fn_parts, fn_whole, d, nbytes, args, side_ok, x, tag, what):
    """args = (x, wp_dgrad, cat, dcat, dx, dw, db) pointers' owners; returns nothing (dx / dw / db are filled)."""
    xx, wp, cat, dcat, dx, dw, db = args
    if not (side_ok and SIDE_DECONV and not _Side.paused):
        ws = WORKSPACE.get(nbytes, xx.device)
        check(fn_whole(ctypes.byref(d), ptr(xx), ptr(wp), ptr(cat), ptr(dcat), ptr(dx), ptr(dw), ptr(db), ptr(ws), nbytes,
                       stream_ptr()), what)
        return
    ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=xx.device)
    check(fn_parts(ctypes.byref(d), ptr(xx), ptr(wp), ptr(cat), ptr(dcat), ptr(dx), ptr(dw), ptr(db), ptr(ws), nbytes, 1,
                   stream_ptr()), what)
    if _Side.stream is None:
        _Side.stream = torch.cuda.Stream(device=xx.device)
    side = _Side.stream
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        check(fn_parts(ctypes.byref(d), ptr(xx), ptr(wp), ptr(cat), ptr(dcat), ptr(dx), ptr(dw), ptr(db), ptr(ws), nbytes, 2,
                       stream_ptr()), what)
    ws.record_stream(side)
    xx.record_stream(side)
    if not _Side.pending:
        _Side.pending = True
        torch.autograd.Variable._execution_engine.queue_callback(side_join)


def deconv2x2_bwd(x, wp_dgrad, cat, dcat, coff, cout, bf16=False, out_w=None, out_b=None):
    n, h, w, cin = x.shape
    prec = precision_of(bf16)
    assert x.dtype == storage_dtype(prec) and cat.dtype == x.dtype and dcat.dtype == x.dtype
    d = DeconvDesc(n, h, w, cin, cout, _pix_stride(cat), coff, prec)
    assert _pix_stride(dcat) == _pix_stride(cat)
    nbytes = _abi.lib().unetk_deconv2x2_bwd_ws_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise _abi.UnetkError("deconv2x2_bwd: unsupported shape Cin={} Cout={}".format(cin, cout))
    dx = torch.empty_like(x)
    dw = out_w if out_w is not None else torch.empty((2, 2, cout, cin), dtype=torch.float32, device=x.device)
    db = out_b if out_b is not None else torch.empty((cout,), dtype=torch.float32, device=x.device)
    assert tuple(dw.shape) == (2, 2, cout, cin) and dw.is_contiguous()
    side_ok = out_w is not None and DEBUG_CAPTURE is None and side_wgrad_on(bf16)
    with _timed(lambda: "deconv2x2_bwd{}(relu_bwd+pw_gemm<dgrad>+deconv_wgrad)".format({0: "", 1: "<bf16>", 2: "<bf16s>"}[prec]),
                16.0 * n * h * w * cin * cout, "{}x{}x{} {}->{}", (n, h, w, cin, cout)):
        _deconv_bwd_call(_abi.lib().unetk_deconv2x2_bwd_parts, _abi.lib().unetk_deconv2x2_bwd, d, nbytes,
                         (x, wp_dgrad, cat, dcat, dx, dw, db), side_ok, x, None, "deconv2x2_bwd")
    return dx, dw, db


def deconv3d_pack(w):
    """w: TF [kd,2,2,Cout,Cin] with kd in {1, 2}."""
    _require_cuda(w)
    kd, kh, kw, cout, cin = w.shape
    assert kh == 2 and kw == 2 and kd in (1, 2)

    def build():
        wp_f = torch.empty(kd * 4 * cin * cout, dtype=torch.float32, device=w.device)
        wp_d = torch.empty_like(wp_f)
        check(_abi.lib().unetk_deconv3d_pack(ptr(w), kd, cin, cout, ptr(wp_f), ptr(wp_d), stream_ptr()), "deconv3d_pack")
        tap = 4 * cin * cout * 4
        items = [(PK_DECONV_F32, 0, cin, cout, w.data_ptr() + a * tap, wp_f.data_ptr() + a * tap, wp_d.data_ptr() + a * tap)
                 for a in range(kd)]
        return wp_f, wp_d, items
    return PACKS.get(w, ("d3d", 0, True), build)


def _pix_stride_nd(t):
    """Pixel stride of a channel-slice view of a dense N...C tensor."""
    assert t.stride(-1) == 1
    return t.stride(-2)


def deconv3d_fwd(x, wp_fwd, bias, cat, coff, cout, kd):
    n, dd, h, w, cin = x.shape
    assert x.is_contiguous()
    d = Deconv3dDesc(n, dd, h, w, cin, cout, kd, _pix_stride_nd(cat), coff)
    with _Timed("deconv3d_fwd", 8.0 * kd * n * dd * h * w * cin * cout, "kd{} {}".format(kd, tuple(x.shape))):
        check(_abi.lib().unetk_deconv3d_fwd(ctypes.byref(d), ptr(x), ptr(wp_fwd), ptr(bias), ptr(cat), stream_ptr()),
              "deconv3d_fwd")
    return cat


def deconv3d_bwd(x, wp_dgrad, cat, dcat, coff, cout, kd, want_dbias, out_w=None):
    n, dd, h, w, cin = x.shape
    d = Deconv3dDesc(n, dd, h, w, cin, cout, kd, _pix_stride_nd(cat), coff)
    assert _pix_stride_nd(dcat) == _pix_stride_nd(cat)
    nbytes = _abi.lib().unetk_deconv3d_bwd_ws_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise _abi.UnetkError("deconv3d_bwd: unsupported shape Cin={} Cout={}".format(cin, cout))
    dx = torch.empty_like(x)
    dw = out_w if out_w is not None else torch.empty((kd, 2, 2, cout, cin), dtype=torch.float32, device=x.device)
    assert tuple(dw.shape) == (kd, 2, 2, cout, cin) and dw.is_contiguous()
    db = torch.empty((cout,), dtype=torch.float32, device=x.device) if want_dbias else None
    side_ok = out_w is not None and DEBUG_CAPTURE is None and _Side.enabled and SIDE_WGRAD3D_VOXELS > 0
    with _Timed("deconv3d_bwd", 16.0 * kd * n * dd * h * w * cin * cout, "kd{} {}".format(kd, tuple(x.shape))):
        _deconv_bwd_call(_abi.lib().unetk_deconv3d_bwd_parts, _abi.lib().unetk_deconv3d_bwd, d, nbytes,
                         (x, wp_dgrad, cat, dcat, dx, dw, db), side_ok, x, None, "deconv3d_bwd")
    return dx, dw, db


def head_desc(n, hw, c, ncls, weight_mode="none", numeric_w=None, proportion_decay=0.0):
    mode = {"none": _abi.W_NONE, "numerical": _abi.W_NUMERICAL, "proportion": _abi.W_PROPORTION,
            "pixelmap": _abi.W_PIXELMAP}[weight_mode]
    d = HeadDesc()
    d.N, d.HW, d.C, d.ncls, d.weight_mode = n, hw, c, ncls, mode
    if mode == _abi.W_NUMERICAL:
        if numeric_w is None or len(numeric_w) != ncls:
            raise KeyError("w_type `numerical` need keyword argument `numeric_w`")
        for i, v in enumerate(numeric_w):
            d.numeric_w[i] = float(v)
    d.proportion_decay = float(proportion_decay or 0.0)
    return d


def head_fwd(d, z, w, b, labels, pixel_w=None, want_probs=False):
    _require_cuda(z, w, b)
    npix = d.N * d.HW
    dev = z.device
    assert z.is_contiguous()
    d.storage = _storage_of(z)
    logits = torch.empty((npix, d.ncls), dtype=torch.float32, device=dev)
    probs = torch.empty_like(logits) if want_probs else None
    nres = _abi.lib().unetk_head_result_floats(ctypes.byref(d))
    nbytes = _abi.lib().unetk_head_ws_bytes(ctypes.byref(d))
    if nres == 0 or nbytes == 0:
        raise _abi.UnetkError("head: bad descriptor")
    result = torch.zeros((nres,), dtype=torch.float32, device=dev)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)    # kept for backward (weight tables)
    with _timed_hbm("head_fwd", z, 1, npix * (4 + 4 * d.ncls * (2 if want_probs else 1))):
        check(_abi.lib().unetk_head_fwd(ctypes.byref(d), ptr(z), ptr(w), ptr(b), ptr(labels), ptr(pixel_w), ptr(logits),
                                        ptr(probs), ptr(result), ptr(ws), nbytes, stream_ptr()), "head_fwd")
    return logits, probs, result, ws


def head_bwd(d, z, w, labels, pixel_w, logits, result, ws, xent_scale, dice_scale, dev_scales=None, out_w=None, out_b=None):
    d.storage = _storage_of(z)
    dz = torch.empty_like(z)
    dw = out_w if out_w is not None else torch.empty((d.C, d.ncls), dtype=torch.float32, device=z.device)
    db = out_b if out_b is not None else torch.empty((d.ncls,), dtype=torch.float32, device=z.device)
    assert tuple(dw.shape) == (d.C, d.ncls) and dw.is_contiguous()
    with _timed_hbm("head_bwd", z, 2, d.N * d.HW * (4 + 4 * d.ncls)):
        check(_abi.lib().unetk_head_bwd(ctypes.byref(d), ptr(z), ptr(w), ptr(labels), ptr(pixel_w), ptr(logits), ptr(result),
                                        float(xent_scale), float(dice_scale), ptr(dev_scales), ptr(dz), ptr(dw), ptr(db),
                                        ptr(ws), ws.numel(), stream_ptr()), "head_bwd")
    return dz, dw, db


def head_predict(probs, ncls, want_preds=True):
    npix = probs.numel() // ncls
    amax = torch.empty((npix,), dtype=torch.uint8, device=probs.device)
    preds = torch.empty((ncls - 1, npix), dtype=torch.uint8, device=probs.device) if want_preds else None
    check(_abi.lib().unetk_head_predict(ptr(probs), npix, ncls, ptr(amax), ptr(preds), stream_ptr()), "head_predict")
    return amax, preds


def boundary_weights(labels):
    """--loss_weight_type boundary (loss_metrics.py:149-165): labels int32 [N,H,W] -> normalised weight map f32."""
    _require_cuda(labels)
    if labels.dim() != 3:
        raise ValueError("loss_weight_type `boundary` is defined for [bs, H, W] labels only (loss_metrics.py:150-151)")
    labels = labels.to(torch.int32).contiguous()
    n, h, w = labels.shape
    wmap = torch.empty((n, h, w), dtype=torch.float32, device=labels.device)
    nbytes = _abi.lib().unetk_boundary_weights_ws_bytes(n, h, w)
    ws = WORKSPACE.get(nbytes, labels.device)
    check(_abi.lib().unetk_boundary_weights(ptr(labels), n, h, w, ptr(wmap), ptr(ws), nbytes, stream_ptr()),
          "boundary_weights")
    return wmap


def lits_batch(slices, seg_slices, sample_tab, clip, out_hw, channels, lab_scale=64, noise_scale=0.0, seed=0):
    """One training batch from device-resident decoded slices (input_pipeline.py:243-284): slices uint16 / seg_slices
    uint8 [n, src_h, src_w] (stored as int16 / uint8 tensors), sample_tab int32 [N, C+7], clip f32 [N, 2]."""
    _require_cuda(slices, seg_slices, sample_tab, clip)
    n = sample_tab.shape[0]
    h, w = out_hw
    assert slices.dtype in (torch.int16, torch.uint16) and seg_slices.dtype == torch.uint8
    assert sample_tab.dtype == torch.int32 and sample_tab.shape[1] == channels + 7 and sample_tab.is_contiguous()
    assert clip.dtype == torch.float32 and tuple(clip.shape) == (n, 2) and clip.is_contiguous()
    assert slices.is_contiguous() and seg_slices.is_contiguous() and slices.shape == seg_slices.shape
    d = _abi.LitsDesc(n, h, w, channels, slices.shape[0], slices.shape[1], slices.shape[2], int(lab_scale),
                      int(seed) & 0xffffffff, float(noise_scale))
    images = torch.empty((n, h, w, channels), dtype=torch.float32, device=slices.device)
    labels = torch.empty((n, h, w), dtype=torch.int32, device=slices.device)
    check(_abi.lib().unetk_lits_batch(ctypes.byref(d), ptr(slices), ptr(seg_slices), ptr(sample_tab), ptr(clip), ptr(images),
                                      ptr(labels), stream_ptr()), "lits_batch")
    return images, labels


def adam_step(p, g, m, v, lr_t, beta1, beta2, eps, gscale=1.0, l2=0.0, decoupled_wd=0.0):
    with _timed_hbm("adam_step", p, 7):            # reads p, g, m, v; writes p, m, v
        check(_abi.lib().unetk_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr_t, beta1, beta2, eps, gscale, l2,
                                         decoupled_wd, stream_ptr()), "adam_step")
    bump_param_gen()


def momentum_step(p, g, acc, lr, mom, nesterov, gscale=1.0, l2=0.0):
    check(_abi.lib().unetk_momentum_step(ptr(p), ptr(g), ptr(acc), p.numel(), lr, mom, 1 if nesterov else 0, gscale,
                                         l2, stream_ptr()), "momentum_step")
    bump_param_gen()


def sumsq(p):
    out = torch.empty((1,), dtype=torch.float32, device=p.device)
    ws = WORKSPACE.get(8192, p.device)
    check(_abi.lib().unetk_sumsq(ptr(p), p.numel(), ptr(out), ptr(ws), 8192, stream_ptr()), "sumsq")
    return out


def png_unfilter(filtered, h, w, bit_depth, out, status):
    """filtered uint8 [n, h * (1 + w * bit_depth / 8)] (inflated PNG rows) -> out [n, h, w] (uint8, or int16 / uint16 storage
    for 16-bit samples) on the device; status int32[1] collects invalid filter types (unetk_png_unfilter)."""
    _require_cuda(filtered, out, status)
    n = filtered.shape[0]
    assert filtered.dtype == torch.uint8 and filtered.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (n, h, w)
    assert out.element_size() == bit_depth // 8 and filtered.shape[1] >= h * (1 + w * bit_depth // 8)
    check(_abi.lib().unetk_png_unfilter(ptr(filtered), filtered.stride(0), n, h, w, bit_depth, ptr(out), h * w, ptr(status),
                                        stream_ptr()), "png_unfilter")
    return out


def nan_watch(value, flag, step):
    """flag (int32[2], zeroed by the caller) becomes (1, step) at the first step whose `value` (a device scalar) is NaN."""
    _require_cuda(value, flag)
    assert value.dtype == torch.float32 and flag.dtype == torch.int32 and flag.numel() >= 2
    check(_abi.lib().unetk_nan_watch(ptr(value), ptr(flag), int(step), stream_ptr()), "nan_watch")


def guide_moments(guide, per_sample):
    """[groups, G + G*G]: E[g_i] and E[g_i g_j] of a (pooled) guide [N, H, W, G] per statistics group (GUNet --fix)."""
    _require_cuda(guide)
    guide = guide.to(torch.float32).contiguous()
    n, g = guide.shape[0], guide.shape[-1]
    hw = guide.numel() // (n * g)
    out = torch.empty((n if per_sample else 1, g + g * g), dtype=torch.float32, device=guide.device)
    check(_abi.lib().unetk_guide_moments(ptr(guide), n, hw, g, 1 if per_sample else 0, ptr(out), stream_ptr()), "guide_moments")
    return out


def norm_drop_pool(d, y, aff):
    """--use_se with --dropout: (sum_p m_p xhat_p, sum_p m_p) per (sample, channel), [2, N, C] (unetk_norm_drop_pool)."""
    d.storage = _storage_of(y)
    sums = torch.empty((2, d.N, d.C), dtype=torch.float32, device=y.device)
    check(_abi.lib().unetk_norm_drop_pool(ctypes.byref(d), ptr(y), ptr(aff[0]), ptr(aff[1]), ptr(sums), stream_ptr()),
          "norm_drop_pool")
    return sums


def norm_se_bwd_add_drop(d, y, dy, aff, e_mat, k1, k2):
    check(_abi.lib().unetk_norm_se_bwd_add_drop(ctypes.byref(d), ptr(y), ptr(dy), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]),
                                                ptr(e_mat.contiguous()), ptr(k1.contiguous()), ptr(k2.contiguous()),
                                                stream_ptr()), "norm_se_bwd_add_drop")
    return dy


def norm_se_bwd_add(d, y, dy, aff, a_mat, k2):
    check(_abi.lib().unetk_norm_se_bwd_add(ctypes.byref(d), ptr(y), ptr(dy), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]),
                                           ptr(a_mat.contiguous()), ptr(k2.contiguous()), stream_ptr()), "norm_se_bwd_add")
    return dy


# ----------------------------------------------------------------------------- autograd nodes
class NormSpec(object):
    """How a conv unit is normalised (the slim arg_scope state around slim.conv2d)."""

    def __init__(self, kind="batch_norm", eps=1e-3, decay=0.999, training=True, bf16=False):
        assert kind in ("batch_norm", "instance_norm", "none")   # "none" = --without_norm: conv + bias + ReLU
        self.kind, self.eps, self.decay, self.training = kind, eps, decay, training
        self.guide_leaky = False   # LGNet: leaky-ReLU on the guide branch before it is added
        self.guide_alpha = 0.2     # its slope (tf.nn.leaky_relu default); 0 = the ReLU of GUNet --fix
        self.guide_per_sample = False   # gw [N, g, C] / gb [N, C]: per-sample folded guide weights (--fix under instance norm)
        self.guide_post = False    # gb is the block [bias, slope for s > 0, slope for s <= 0, post-shift] x C (after_affine + --fix)
        self.dropout = None        # (keep_prob, seed): slim.dropout on the normalised value (GUNet --dropout), training only
        self.se = None             # GUNet --use_se: callable (pooled [N, C], context slice) -> gains [N, C] (torch graph)
        self.bf16 = precision_of(bf16)   # 0 fp32 | 1 UNETK_BF16 (bf16 operands, fp32 tensors) | 2 UNETK_BF16S (+ bf16 tensors)

    @property
    def per_sample(self):
        return self.kind == "instance_norm"


class Conv3x3NormRelu(_Op):
    """z = relu(norm(conv3x3(x, w)) [+ spatial guide modulation]) -- one slim.conv2d(x, C, 3) unit:
    conv without bias, slim.batch_norm / slim.instance_norm with optional centre (beta) and scale (gamma),
    ReLU (NetworksV2/UNet.py:41-56,79; GUNet.py:162-217 `modulated_conv_block`)."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, moving_mean, moving_var, spec, out, guide, gw, gb, den=None, dilation=1, se_feat=None):
        _require_cuda(x, w)
        cin, cout = w.shape[2], w.shape[3]
        dilation = int(dilation or 1)
        if dilation != 1 and (bool(getattr(spec, "bf16", False)) or not conv_uses_mfma(cin, cout)):
            raise _abi.UnetkError("atrous conv (rate {}) needs the fp32 MFMA path (Cin % 16, Cout % 64)".format(dilation))
        if den is not None:
            den = den.contiguous()
        mfma = conv_uses_mfma(cin, cout)
        prec = precision_of(getattr(spec, "bf16", 0))
        bf16 = prec if conv_uses_bf16(cin, cout) else 0
        if prec == _abi.BF16S:
            # bf16 storage: the matrix kernels need 64-channel blocks on both sides; the first layer (small Cin, direct
            # kernel) reads the fp32 image and writes bf16
            if mfma and (cin % 64 or cout % 64):
                raise _abi.UnetkError("bf16 storage needs Cin % 64 == 0 and Cout % 64 == 0 (got {}->{})".format(cin, cout))
            bf16 = prec
        need_dx = ctx.needs_input_grad[0]
        if mfma:
            wp_f, wp_d = conv3x3_pack(w, want_dgrad=need_dx, bf16=bf16)
        else:
            wp_f, wp_d = w, None
            if need_dx:
                raise _abi.UnetkError("conv3x3 input gradient needs Cin%64==0 and Cout%16==0 "
                                      "(got {}->{})".format(cin, cout))
        plain = spec.kind == "none"
        se = getattr(spec, "se", None)
        use_batch_stats = (spec.training or spec.per_sample) and not plain
        if se is not None and (plain or den is not None):
            raise _abi.UnetkError("--use_se needs a normalised unit and no other density gain")
        # ---- inference fast path: the affine is known before the conv (moving statistics / --without_norm), so conv +
        # (scale, shift) + ReLU [+ the 2 x 2 max-pool the unit feeds] are ONE kernel and the raw output never exists
        fprec = precision_of(bf16)          # UNETK_BF16S: bf16 activations in and out (the first layer reads the fp32 image)
        if FUSE_EVAL and not spec.training and not use_batch_stats and se is None and den is None and guide is None \
                and gb is None and fprec in (_abi.FP32, _abi.BF16S) and dilation == 1:
            n_, h_, w_ = x.shape[0], x.shape[1], x.shape[2]
            z = out if out is not None else torch.empty((n_, h_, w_, cout), dtype=storage_dtype(fprec), device=x.device)
            fd = ConvDesc(n_, h_, w_, cin, cout, _pix_stride(x), _pix_stride(z), fprec, 1)
            want_pool = bool(getattr(ctx, "want_pool", False)) and POOL_FUSED
            with_pool = want_pool and _abi.lib().unetk_conv3x3_fwd_affine_ok(ctypes.byref(fd), 1) == 1
            if (with_pool or _abi.lib().unetk_conv3x3_fwd_affine_ok(ctypes.byref(fd), 0) == 1) and z.dtype == storage_dtype(fprec):
                nd = norm_desc((n_, h_, w_, cout), False, _pix_stride(z), 0, 0, 0)
                if plain:
                    sc, sh = torch.ones_like(beta), beta.detach().contiguous()
                else:
                    aff = norm_finalize(nd, None, 0, gamma, beta, spec.eps, spec.decay, False, moving_mean, moving_var, x.device)
                    sc, sh = aff[2], aff[3]
                pooled = torch.empty((n_, h_ // 2, w_ // 2, cout), dtype=z.dtype, device=x.device) if with_pool else None
                tag = _igemm_tag(cin, cout, bf16, h_, n_, w_) + ("+affine+relu+pool" if with_pool else "+affine+relu")
                nws = _abi.lib().unetk_conv3x3_ws_bytes(ctypes.byref(fd))
                ws = WORKSPACE.get(nws, x.device) if nws else None
                with _timed(tag, 18.0 * n_ * h_ * w_ * cin * cout, "infer {}x{}x{} {}->{}", (n_, h_, w_, cin, cout)):
                    check(_abi.lib().unetk_conv3x3_fwd_affine(ctypes.byref(fd), ptr(x), ptr(wp_f), ptr(sc), ptr(sh), ptr(z),
                                                              ptr(pooled), cout, ptr(ws), nws, stream_ptr()), "conv3x3_fwd_affine")
                ctx.pooled = pooled
                return alias(z) if out is not None else z
        y, stats, rows = conv3x3_fwd(x, wp_f, cout, want_stats=use_batch_stats or se is not None, bf16=bf16, dilation=dilation)
        z = out if out is not None else torch.empty_like(y)
        g_ch = 0 if guide is None else guide.shape[-1]
        if g_ch:
            gw = gw.contiguous()
        if gb is not None:                  # without a guide: a bare per-channel shift after the gain (after_affine)
            gb = gb.contiguous()
        d = norm_desc(y.shape, spec.per_sample, _pix_stride(z), g_ch, cout if g_ch else 0, 0)
        if g_ch and getattr(spec, "guide_leaky", False):      # LGNet: u = t + leaky_relu(guide . gw + gb); --fix: ReLU
            alpha = float(getattr(spec, "guide_alpha", 0.2))
            d.guide_leaky = 1 if alpha == 0.2 else 2           # 1 = tf.nn.leaky_relu's default slope, 2 = guide_alpha
            d.guide_alpha = alpha
        if g_ch and getattr(spec, "guide_post", False):       # GUNet after_affine + --fix: gb = [bias, slope+, slope-, post-shift] rows
            d.guide_leaky = 3
        if getattr(spec, "guide_per_sample", False):
            d.guide_per_sample = 1
            d.gw_stride = cout
        drop = getattr(spec, "dropout", None)
        if drop is not None and spec.training:
            d.dropout_keep, d.dropout_seed = float(drop[0]), int(drop[1]) & 0xFFFFFFFF
            if den is None and (g_ch or gb is not None) and not d.guide_leaky:
                den = torch.ones((x.shape[0], cout), dtype=torch.float32, device=x.device)   # the backward's separate sum du
        if plain:
            # --without_norm (UNet.py:47-48): z = relu(y + bias); `beta` carries the conv bias
            d.affine_only = 1
            one, zero = torch.ones_like(beta), torch.zeros_like(beta)
            aff = torch.stack([zero, one, one, beta.detach()]).reshape(4, 1, cout).contiguous()
        else:
            aff = norm_finalize(d, stats, rows, gamma, beta, spec.eps, spec.decay, spec.training, moving_mean,
                                moving_var, y.device)
        se_graph = None
        if se is not None:
            # GUNet.py:192-201: gains = sigmoid(fc(relu(fc(concat(mean_hw(net), context))))).  mean_hw(net)[b, c] =
            # gamma * xhat_mean[b, c] + beta with xhat_mean from the per-sample means of the raw conv output (the same
            # statistic partials, reduced per sample); the tiny [N, C] graph below is ordinary torch autograd, run
            # backwards from inside this node's backward (the gains' gradient is only known there)
            m_mean = None
            if d.dropout_keep > 0:
                # --dropout as well (GUNet.py:189-201): the gate pools the DROPPED-OUT value, mean_hw(m (gamma xhat + beta)) =
                # gamma * mean(m xhat) + beta * mean(m): two per-sample sums of their own (unetk_norm_drop_pool)
                sums = norm_drop_pool(d, y, aff)
                xhat_mean, m_mean = (sums[0] / float(d.HW)).detach(), (sums[1] / float(d.HW)).detach()
            else:
                ps = norm_desc(y.shape, True)
                mean_b = norm_finalize(ps, stats, rows, None, None, spec.eps, 0.0, True, None, None, y.device)[0]    # [N, C]
                xhat_mean = ((mean_b - aff[0]) * aff[1]).detach()
            with torch.enable_grad():
                g_leaf = gamma.detach().requires_grad_(True) if gamma is not None else None
                b_leaf = beta.detach().requires_grad_(True) if beta is not None else None
                pooled = xhat_mean if g_leaf is None else xhat_mean * g_leaf
                if b_leaf is not None:
                    pooled = pooled + (b_leaf if m_mean is None else m_mean * b_leaf)
                pooled = pooled + torch.zeros_like(xhat_mean).requires_grad_(True) if not pooled.requires_grad else pooled
                pooled.retain_grad()
                # the context slice comes from the OUTER graph (the context MLP): inside, a detached leaf stands in for it
                # and its gradient is handed back as this node's gradient for `se_feat`
                f_leaf = se_feat.detach().requires_grad_(True) if se_feat is not None else None
                gains = se(pooled, f_leaf)
            den = gains.detach().contiguous()
            se_graph = (g_leaf, b_leaf, pooled, gains, xhat_mean, f_leaf, m_mean)
        ctx.pooled = None
        if getattr(ctx, "want_pool", False) and POOL_FUSED and se is None and den is None and g_ch == 0 and gb is None \
                and d.dropout_keep == 0 and y.shape[1] % 2 == 0 and y.shape[2] % 2 == 0:
            ctx.pooled = norm_apply_relu_pool(d, y, aff, z)        # Conv3x3NormReluPool: the pool rides on the apply pass
        else:
            norm_apply_relu(d, y, aff, z, guide, gw, gb, den)
        # a unit another conv unit may fuse its input gradient with (see conv3x3_dgrad): plain normalised units only
        simple = (not plain and se is None and den is None and g_ch == 0 and gb is None and d.dropout_keep == 0)
        if spec.training:
            src = getattr(x, "_unetk_unit", None)
            ctx.producer = src if (need_dx and src is not None and dilation == 1) else None
            ctx.simple = simple
            if getattr(ctx, "want_pool", False):
                ctx.unit_saved = (x, y, aff, guide, gw, gb, den)       # Conv3x3NormReluPool saves these together with its outputs
            else:
                ctx.save_for_backward(x, y, aff, guide, gw, gb, den)
            ctx.se_graph = se_graph
            ctx.wp_d = wp_d
            ctx.need_dx = need_dx
            ctx.bf16 = bf16
            ctx.dilation = dilation
            ctx.desc = d
            ctx.has = (gamma is not None, beta is not None)
            # gamma / beta of a unit with an SE gate also get gradient from the gate's graph: those stay with autograd
            ctx.sinks = (grad_sink(w, ctx), grad_sink(gamma, ctx) if se is None else None, grad_sink(beta, ctx) if se is None and not plain else None)
            ctx.w_dbg = w.detach() if DEBUG_CAPTURE is not None else None
            ctx.gb_dbg = (gamma, beta) if DEBUG_CAPTURE is not None else None
            ctx.z_dbg = z.detach() if DEBUG_CAPTURE is not None else None      # the forward's own ReLU mask, for the checkers
        zr = alias(z) if out is not None else z
        if spec.training and simple and out is None:
            zr._unetk_unit = (y, aff, spec.per_sample)
        return zr

    @staticmethod
    def backward(ctx, dz):
        return Conv3x3NormRelu._backward(ctx, dz)

    @staticmethod
    def _backward(ctx, dz, pool=None):
        """`pool` = (dp, dskip): the unit's activation fed max_pool2d and the skip connection (Conv3x3NormReluPool) and its
        gradient dz = dskip + route(dp) is formed inside the norm backward's two passes instead of being written first."""
        x, y, aff, guide, gw, gb, den = ctx.saved_tensors[:7]
        pre = None
        if pool is None:
            pre = FUSED_NBR.pop(dz.data_ptr(), None) if DEBUG_CAPTURE is None else None
            if pre is not None and not (ctx.simple and pre[0] == y.data_ptr() and pre[1] == tuple(dz.shape) and dz.is_contiguous()
                                        and pre[4] == dz._version):
                pre = None
            if dz.stride(3) != 1:
                dz = dz.contiguous()
            if dz.dtype != y.dtype:
                raise _abi.UnetkError("gradient dtype {} does not match the stored activations ({})".format(dz.dtype, y.dtype))
        dden = dfeat = None
        debug = DEBUG_CAPTURE is not None            # the checkers read the returned tensors
        sw, sg, sb = (None, None, None) if debug else (_take(ctx.sinks[0]), _take(ctx.sinks[1]), _take(ctx.sinks[2]))
        if pool is not None:
            dy, dgamma, dbeta = norm_relu_bwd_pool(ctx.desc, y, pool[1], pool[0], aff, ctx.has[0], ctx.has[1],
                                                   out_gamma=sg, out_beta=sb)
            dgw = dgb = None
        elif den is None:
            dy, dgamma, dbeta, dgw, dgb = norm_relu_bwd(ctx.desc, y, dz, aff, ctx.has[0], ctx.has[1], guide, gw, gb,
                                                        out_gamma=sg, out_beta=sb, pre=(pre[2], pre[3]) if pre else None)
        else:
            dy, dgamma, dbeta, dgw, dgb, dden = norm_relu_bwd(ctx.desc, y, dz, aff, ctx.has[0], ctx.has[1], guide, gw, gb,
                                                              den, out_gamma=sg, out_beta=sb)
        if getattr(ctx, "se_graph", None) is not None:
            g_leaf, b_leaf, pooled, gains, xhat_mean, f_leaf, m_mean = ctx.se_graph
            torch.autograd.backward(gains, dden)              # FC parameters accumulate here; pooled.grad = d loss / d pooled
            gp = pooled.grad
            dfeat = f_leaf.grad if f_leaf is not None else None
            if g_leaf is not None and g_leaf.grad is not None:
                dgamma = dgamma + g_leaf.grad if dgamma is not None else g_leaf.grad
            if b_leaf is not None and b_leaf.grad is not None:
                dbeta = dbeta + b_leaf.grad if dbeta is not None else b_leaf.grad
            if m_mean is not None:
                # with dropout the pooled value depends on y under either norm: dt gets m_p * E[b, c]; the statistics terms of
                # the (linear) norm backward follow from the forward's two sums: mean(m E) = E mean(m), mean(m E xhat) = E mean(m xhat)
                e_mat = gp / float(ctx.desc.HW)
                k1, k2 = e_mat * m_mean, e_mat * xhat_mean
                if not ctx.desc.per_sample:
                    k1, k2 = k1.mean(dim=0, keepdim=True), k2.mean(dim=0, keepdim=True)
                norm_se_bwd_add_drop(ctx.desc, y, dy, aff, e_mat, k1, k2)
            elif not ctx.desc.per_sample:                     # under instance norm the pooled value does not depend on y
                hw = float(ctx.desc.HW)
                a_mat = gp / hw
                k2 = (a_mat * xhat_mean).mean(dim=0, keepdim=True)
                a_mat = a_mat - a_mat.mean(dim=0, keepdim=True)
                norm_se_bwd_add(ctx.desc, y, dy, aff, a_mat, k2)
            dden = None                                       # the gains are internal to this node
            ctx.se_graph = None
        if ctx.desc.dropout_keep > 0 and den is not None and not ctx.needs_input_grad[11]:
            dden = None
        on_side = side_wgrad_on(ctx.bf16) and not debug and sw is not None and ctx.need_dx
        if not on_side:
            dw = conv3x3_wgrad(x, dy, bf16=ctx.bf16, dilation=ctx.dilation, out=sw)
        elif SIDE_WGRAD_FIRST:      # queued behind dy only: runs beside this unit's input gradient
            dw = _wgrad_on_side(x, dy, ctx.bf16, ctx.dilation, sw)
        dx = conv3x3_dgrad(dy, ctx.wp_d, x.shape[3], bf16=ctx.bf16, dilation=ctx.dilation,
                           producer=ctx.producer if DEBUG_CAPTURE is None else None) if ctx.need_dx else None
        if on_side and not SIDE_WGRAD_FIRST:      # behind the input gradient in issue order: beside the next unit's norm backward
            dw = _wgrad_on_side(x, dy, ctx.bf16, ctx.dilation, sw)
        if DEBUG_CAPTURE is not None:
            DEBUG_CAPTURE.append(dict(x=x, y=y, dilation=ctx.dilation, z=ctx.z_dbg, gamma=ctx.gb_dbg[0], beta=ctx.gb_dbg[1], aff=aff, dz=dz, dy=dy, dw=dw,
                                      dx=dx, dgamma=dgamma, dbeta=dbeta, w=ctx.w_dbg, guide=guide, gw=gw, gb=gb,
                                      dgw=dgw, dgb=dgb, per_sample=bool(ctx.desc.per_sample), bf16=ctx.bf16, den=den,
                                      dden=dden, guide_leaky=bool(ctx.desc.guide_leaky), plain=bool(ctx.desc.affine_only)))
        return dx, _ret(dw, sw), _ret(dgamma, sg), _ret(dbeta, sb), None, None, None, None, None, dgw, dgb, dden, None, dfeat


POOL_FUSED = os.environ.get("UNETK_POOL_FUSED", "1") != "0"     # 0: separate pool-backward pass (measurement)


class Conv3x3NormReluPool(_Op):
    """(p, z) = (max_pool2d(z, 2, 2), z) with z = Conv3x3NormRelu(x, ...): the second conv of an encoder level, whose
    activation feeds the pool and -- through the concat buffer it was written into -- the skip connection
    (NetworksV2/UNet.py:79-81,93).  One node for conv unit + pool so that the backward gets BOTH gradients of z, dp and
    dskip, and never writes their sum: unetk_norm_relu_bwd_pool routes dp to each window's first maximum and adds dskip
    inside the norm backward's reduction and apply passes (the MaxPoolSkip node's pass over z / dz disappears).  Plain
    normalised units only (what UNet's encoder is); anything else takes the two separate passes."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, moving_mean, moving_var, spec, out):
        ctx.want_pool = True
        z = Conv3x3NormRelu.forward(ctx, x, w, gamma, beta, moving_mean, moving_var, spec, out, None, None, None)
        p = ctx.pooled if ctx.pooled is not None else maxpool2_fwd(z)
        ctx.pooled = None
        if spec.training:
            # the node's own outputs go through save_for_backward like its inputs (a plain attribute would tie
            # p -> grad_fn -> ctx -> p into a cycle that only a backward run breaks); both are alive anyway: z is the skip
            # inside the concat buffer, p the next unit's input
            ctx.save_for_backward(*(ctx.unit_saved + (z, p)))
            ctx.unit_saved = None
        return p, z

    @staticmethod
    def backward(ctx, dp, dskip):
        z, p = ctx.saved_tensors[7:9]
        if dskip is not None and dskip.stride(3) != 1:
            dskip = dskip.contiguous()
        plain_unit = ctx.saved_tensors[3] is None and ctx.saved_tensors[5] is None and ctx.saved_tensors[6] is None \
            and ctx.desc.dropout_keep == 0 and getattr(ctx, "se_graph", None) is None
        if POOL_FUSED and DEBUG_CAPTURE is None and plain_unit and dp is not None and dskip is not None \
                and dp.dtype == z.dtype and dskip.dtype == z.dtype:
            return Conv3x3NormRelu._backward(ctx, None, pool=(dp.contiguous(), dskip))[:8]
        dz = dskip if dp is None else maxpool2_bwd(z, p, dp, dskip)
        return Conv3x3NormRelu._backward(ctx, dz)[:8]


class FullyConnected(_Op):
    """y = dropout(act(x . w + b)), act = identity (relu 0 / False), ReLU (1 / True) or sigmoid (2: the SE gate,
    GUNet.py:199) -- one slim.fully_connected (+ slim.dropout) of GUNet's context MLP
    (NetworksV2/Backbone/slim_nets.py:43-56; GUNet.py:50-60).  w is TF's [in, out].  keep_prob None = no dropout;
    the mask (0 or 1/keep_prob per element, counter RNG keyed by `seed`) is kept for the backward."""

    @staticmethod
    def forward(ctx, x, w, b, relu, keep_prob, seed):
        _require_cuda(x, w)
        x = x.contiguous()
        bsz, k = x.shape
        n = w.shape[1]
        assert w.shape[0] == k and w.is_contiguous()
        y = torch.empty((bsz, n), dtype=torch.float32, device=x.device)
        mask = torch.empty_like(y) if keep_prob is not None else None
        check(_abi.lib().unetk_fc_fwd(ptr(x), ptr(w), ptr(b), ptr(y), ptr(mask), bsz, k, n, int(relu or 0),
                                      float(keep_prob) if keep_prob is not None else 1.0, int(seed) & 0xFFFFFFFF,
                                      stream_ptr()), "fc_fwd")
        ctx.save_for_backward(x, w, y, mask)
        ctx.relu = relu
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, mask = ctx.saved_tensors
        dy = dy.contiguous()
        bsz, k = x.shape
        n = w.shape[1]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty((n,), dtype=torch.float32, device=x.device) if ctx.has_b else None
        ws = torch.empty_like(y)
        check(_abi.lib().unetk_fc_bwd(ptr(x), ptr(w), ptr(y), ptr(mask), ptr(dy), ptr(dx), ptr(dw), ptr(db), ptr(ws), bsz, k,
                                      n, int(ctx.relu or 0), stream_ptr()), "fc_bwd")
        return dx, dw, db, None, None, None


class Conv1d(_Op):
    """y = relu(conv1d(x, w) + b): slim.conv1d(x, C, k) with slim's defaults (stride 1, SAME, bias, ReLU) -- the layers of
    GUNet's 1-D VGG context models (NetworksV2/Backbone/slim_nets.py:60-144).  x [B, L, Cin], w TF [k, Cin, Cout]."""

    @staticmethod
    def forward(ctx, x, w, b, relu=True):
        _require_cuda(x, w)
        x = x.contiguous()
        bsz, length, cin = x.shape
        k, cin_w, cout = w.shape
        assert cin_w == cin and w.is_contiguous() and k in (1, 3)
        y = torch.empty((bsz, length, cout), dtype=torch.float32, device=x.device)
        check(_abi.lib().unetk_conv1d_fwd(ptr(x), ptr(w), ptr(b), ptr(y), bsz, length, cin, cout, k, 1 if relu else 0,
                                          stream_ptr()), "conv1d_fwd")
        ctx.save_for_backward(x, w, y)
        ctx.relu, ctx.has_b = bool(relu), b is not None
        ctx.sinks = (grad_sink(w, ctx), grad_sink(b, ctx))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        bsz, length, cin = x.shape
        k, _, cout = w.shape
        sw, sb = _take(ctx.sinks[0]), _take(ctx.sinks[1])
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = sw if sw is not None else torch.empty_like(w)
        db = (sb if sb is not None else torch.empty((cout,), dtype=torch.float32, device=x.device)) if ctx.has_b else None
        ws = torch.empty_like(y)
        check(_abi.lib().unetk_conv1d_bwd(ptr(x), ptr(w), ptr(y), ptr(dy), ptr(dx), ptr(dw), ptr(db), ptr(ws), bsz, length,
                                          cin, cout, k, 1 if ctx.relu else 0, stream_ptr()), "conv1d_bwd")
        return dx, _ret(dw, sw), (_ret(db, sb) if ctx.has_b else None), None


class MaxPool1d(_Op):
    """tf.layers.max_pooling1d(x, 2, 2, padding="same") on [B, L, C] (slim_nets.py:73 ...): Lo = ceil(L / 2)."""

    @staticmethod
    def forward(ctx, x):
        _require_cuda(x)
        x = x.contiguous()
        bsz, length, c = x.shape
        y = torch.empty((bsz, (length + 1) // 2, c), dtype=torch.float32, device=x.device)
        check(_abi.lib().unetk_maxpool1d_fwd(ptr(x), ptr(y), bsz, length, c, stream_ptr()), "maxpool1d_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        bsz, length, c = x.shape
        dx = torch.empty_like(x)
        check(_abi.lib().unetk_maxpool1d_bwd(ptr(x), ptr(dy.contiguous()), ptr(dx), bsz, length, c, stream_ptr()),
              "maxpool1d_bwd")
        return dx


class SpatialMean(_Op):
    """tf.reduce_mean(x, axis=(1, 2)) on NHWC (GUNet.py:108, the conv context subnet)."""

    @staticmethod
    def forward(ctx, x):
        _require_cuda(x)
        x = x.contiguous()
        n, c = x.shape[0], x.shape[-1]
        hw = x.numel() // (n * c)
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        check(_abi.lib().unetk_spatial_mean_fwd(ptr(x), ptr(y), n, hw, c, stream_ptr()), "spatial_mean_fwd")
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c = ctx.shape[0], ctx.shape[-1]
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        check(_abi.lib().unetk_spatial_mean_bwd(ptr(dy.contiguous()), ptr(dx), n, dx.numel() // (n * c), c, stream_ptr()),
              "spatial_mean_bwd")
        return dx


class Conv3dNormRelu(_Op):
    """z = relu(norm(conv3d(x, w))) -- one slim.conv3d unit of UNet3D (UNet3D.py:108-121,153,165): kernel
    (1,3,3) or (3,3,3), stride 1 / (1,2,2) / (2,2,2), SAME, no bias, instance or batch norm, ReLU."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, moving_mean, moving_var, spec, stride, out):
        _require_cuda(x, w)
        kd, cin, cout = w.shape[0], w.shape[3], w.shape[4]
        mfma = conv_uses_mfma(cin, cout)
        need_dx = ctx.needs_input_grad[0]
        live8 = getattr(w, "unetk_live8", None)      # set on channel-padded filters by PaddedParamStore
        d = conv3d_desc(x.shape, cout, kd, stride, x_stride=_pix_stride_nd(x), live8=live8)
        if mfma:
            wp_f, wp_d = conv3d_pack(w, want_dgrad=need_dx)
        else:
            if kd != 1 or need_dx:
                raise _abi.UnetkError("conv3d with Cin={} Cout={} needs the MFMA path (Cin%16, Cout%32)".format(cin, cout))
            wp_f, wp_d = w, None
        plain = spec.kind == "none"                # --without_norm: z = relu(y + bias); `beta` carries the conv bias
        use_batch_stats = (spec.training or spec.per_sample) and not plain
        y, stats, rows = conv3d_fwd(x, wp_f, d, want_stats=use_batch_stats)
        z = out if out is not None else torch.empty_like(y)
        nd = norm_desc(y.shape, spec.per_sample, _pix_stride_nd(z))
        if plain:
            nd.affine_only = 1
            one, zero = torch.ones_like(beta), torch.zeros_like(beta)
            aff = torch.stack([zero, one, one, beta.detach()]).reshape(4, 1, cout).contiguous()
        else:
            aff = norm_finalize(nd, stats, rows, gamma, beta, spec.eps, spec.decay, spec.training, moving_mean, moving_var,
                                y.device)
        norm_apply_relu(nd, y, aff, z)
        if spec.training:
            ctx.save_for_backward(x, y, aff)
            ctx.wp_d, ctx.need_dx, ctx.d, ctx.nd, ctx.live8 = wp_d, need_dx, d, nd, live8
            ctx.has = (gamma is not None, beta is not None)
            ctx.sinks = (grad_sink(w, ctx), grad_sink(gamma, ctx), grad_sink(beta, ctx) if not plain else None)
            ctx.dbg = (w.detach(), gamma, beta, stride, z.detach()) if DEBUG_CAPTURE is not None else None
        return alias(z) if out is not None else z

    @staticmethod
    def backward(ctx, dz):
        x, y, aff = ctx.saved_tensors
        if dz.stride(-1) != 1:
            dz = dz.contiguous()
        debug = DEBUG_CAPTURE is not None
        sw, sg, sb = (None, None, None) if debug else (_take(ctx.sinks[0]), _take(ctx.sinks[1]), _take(ctx.sinks[2]))
        dy, dgamma, dbeta, _, _ = norm_relu_bwd_nd(ctx.nd, y, dz, aff, ctx.has[0], ctx.has[1], out_gamma=sg, out_beta=sb)
        voxels = dy.numel() // dy.shape[-1]
        on_side = (not debug) and sw is not None and ctx.need_dx and 0 < voxels <= SIDE_WGRAD3D_VOXELS and not _Side.paused
        if not on_side:
            dw = conv3d_wgrad(x, dy, ctx.d, out=sw)
        elif SIDE_WGRAD3D_FIRST:
            dw = _wgrad3d_on_side(x, dy, ctx.d, sw)      # queued behind dy only: runs BESIDE this layer's input gradient
        dense = conv3d_desc(x.shape, ctx.d.Cout, ctx.d.kd, (ctx.d.sd, ctx.d.shw, ctx.d.shw), live8=ctx.live8)   # dx is dense
        dx = conv3d_dgrad(dy, ctx.wp_d, dense) if ctx.need_dx else None
        if on_side and not SIDE_WGRAD3D_FIRST:
            dw = _wgrad3d_on_side(x, dy, ctx.d, sw)      # queued behind the input gradient: runs beside the NEXT unit's norm backward
        if DEBUG_CAPTURE is not None:
            DEBUG_CAPTURE.append(dict(kind="conv3d", x=x, y=y, w=ctx.dbg[0], gamma=ctx.dbg[1], beta=ctx.dbg[2],
                                      stride=ctx.dbg[3], z=ctx.dbg[4], dz=dz, dy=dy, dw=dw, dx=dx, dgamma=dgamma, dbeta=dbeta,
                                      per_sample=bool(ctx.nd.per_sample), plain=bool(ctx.nd.affine_only)))
        return dx, _ret(dw, sw), _ret(dgamma, sg), _ret(dbeta, sb), None, None, None, None, None


class Deconv3dConcat(_Op):
    """cat = concat(skip, relu(conv3d_transpose(x, w, kernel == stride))) -- UNet3D.py:161-163 (no bias);
    `skip` already lives in cat[..., :C], the kernel fills cat[..., C:]."""

    @staticmethod
    def forward(ctx, x, w, skip, cat):
        _require_cuda(x, w, cat)
        kd, cout = w.shape[0], w.shape[3]
        coff = cat.shape[-1] - cout
        assert skip.data_ptr() == cat.data_ptr() and skip.shape[-1] == coff
        wp_f, wp_d = deconv3d_pack(w)
        deconv3d_fwd(x, wp_f, None, cat, coff, cout, kd)
        ctx.save_for_backward(x, cat)
        ctx.wp_d, ctx.cout, ctx.coff, ctx.kd = wp_d, cout, coff, kd
        ctx.sink = grad_sink(w, ctx)
        ctx.w_dbg = w.detach() if DEBUG_CAPTURE is not None else None
        return alias(cat)

    @staticmethod
    def backward(ctx, dcat):
        x, cat = ctx.saved_tensors
        dcat = dcat.contiguous()
        sw = None if DEBUG_CAPTURE is not None else _take(ctx.sink)
        dx, dw, _ = deconv3d_bwd(x, ctx.wp_d, cat, dcat, ctx.coff, ctx.cout, ctx.kd, want_dbias=False, out_w=sw)
        if DEBUG_CAPTURE is not None:
            DEBUG_CAPTURE.append(dict(kind="deconv3d", x=x, w=ctx.w_dbg, cat=cat.clone(), dcat=dcat, dx=dx, dw=dw,
                                      coff=ctx.coff, kd=ctx.kd))
        return dx, _ret(dw, sw), dcat[..., :ctx.coff], None


class MaxPool2x2(_Op):
    @staticmethod
    def forward(ctx, x):
        p = maxpool2_fwd(x)
        ctx.save_for_backward(x, p)
        return p

    @staticmethod
    def backward(ctx, dp):
        x, p = ctx.saved_tensors
        return maxpool2_bwd(x, p, dp)


class MaxPoolSkip(_Op):
    """(p, skip) = (max_pool2d(x), x): the encoder activation feeds both the pool and the skip connection
    (UNet.py:80-81,93).  Returning the skip from the same node lets the backward sum the two gradients inside the pool
    backward kernel (dx = route(dp) + dskip, dskip read in place from the concat buffer's gradient) instead of a
    separate elementwise pass."""

    @staticmethod
    def forward(ctx, x):
        p = maxpool2_fwd(x)
        ctx.save_for_backward(x, p)
        return p, alias(x)

    @staticmethod
    def backward(ctx, dp, dskip):
        x, p = ctx.saved_tensors
        if dskip is not None and dskip.stride(3) != 1:
            dskip = dskip.contiguous()
        if dp is None:
            return dskip
        return maxpool2_bwd(x, p, dp, dskip)


class DeconvConcat(_Op):
    """cat = concat(skip, relu(conv2d_transpose(x, w, k=2, s=2) + b)); `skip` already lives in
    cat[..., :C] (the encoder wrote it there), the kernel fills cat[..., C:] -- zero-copy concat."""

    @staticmethod
    def forward(ctx, x, w, b, skip, cat, bf16=False):
        _require_cuda(x, w, b, cat)
        cout = w.shape[2]
        coff = cat.shape[3] - cout
        assert skip.data_ptr() == cat.data_ptr() and skip.shape[3] == coff
        bf16 = precision_of(bf16) if conv_uses_bf16(w.shape[3], cout) else 0
        if x.dtype == torch.bfloat16 and (bf16 != _abi.BF16S or w.shape[3] % 64 or cout % 64):
            raise _abi.UnetkError("bf16 storage needs Cin % 64 == 0 and Cout % 64 == 0 in the transposed conv")
        wp_f, wp_d = deconv2x2_pack(w, bf16)
        deconv2x2_fwd(x, wp_f, b, cat, coff, cout, bf16)
        ctx.save_for_backward(x, cat)
        ctx.wp_d = wp_d
        ctx.bf16 = bf16
        ctx.cout, ctx.coff = cout, coff
        ctx.has_b = b is not None            # SmallUNet's conv2d_transpose has biases_initializer=None
        ctx.sinks = (grad_sink(w, ctx), grad_sink(b, ctx))
        ctx.wb_dbg = (w.detach(), b.detach() if b is not None else None) if DEBUG_CAPTURE is not None else None
        return alias(cat)

    @staticmethod
    def backward(ctx, dcat):
        x, cat = ctx.saved_tensors
        dcat = dcat.contiguous()
        sw, sb = (None, None) if DEBUG_CAPTURE is not None else (_take(ctx.sinks[0]), _take(ctx.sinks[1]))
        dx, dw, db = deconv2x2_bwd(x, ctx.wp_d, cat, dcat, ctx.coff, ctx.cout, ctx.bf16, out_w=sw, out_b=sb)
        dskip = dcat[..., :ctx.coff]
        if DEBUG_CAPTURE is not None:
            DEBUG_CAPTURE.append(dict(kind="deconv", x=x, w=ctx.wb_dbg[0], b=ctx.wb_dbg[1], cat=cat.clone(),
                                      dcat=dcat, dx=dx, dw=dw, db=db, coff=ctx.coff, bf16=ctx.bf16))
        return dx, _ret(dw, sw), (_ret(db, sb) if ctx.has_b else None), dskip, None, None


class DeconvConcatFront(_Op):
    """cat = concat(relu(conv2d_transpose(x, w, k=2, s=2) [+ b]), skip_a, skip_b) -- InterUNet's decoder (InterUNet.py:150-155:
    the up-sampled tensor FIRST, then the skips of the two encoders).  The skips already live in cat[..., C:] (their
    encoders wrote them there); the kernel fills cat[..., :C]."""

    @staticmethod
    def forward(ctx, x, w, b, skip_a, skip_b, cat):
        _require_cuda(x, w, cat)
        cout = w.shape[2]
        ca, cb = skip_a.shape[3], skip_b.shape[3]
        assert cat.shape[3] == cout + ca + cb
        assert skip_a.data_ptr() == cat.data_ptr() + 4 * cout and skip_b.data_ptr() == cat.data_ptr() + 4 * (cout + ca)
        wp_f, wp_d = deconv2x2_pack(w, False)
        deconv2x2_fwd(x, wp_f, b, cat, 0, cout, False)
        ctx.save_for_backward(x, cat)
        ctx.wp_d, ctx.cout, ctx.ca, ctx.has_b = wp_d, cout, ca, b is not None
        ctx.wb_dbg = (w.detach(), b.detach() if b is not None else None) if DEBUG_CAPTURE is not None else None
        return alias(cat)

    @staticmethod
    def backward(ctx, dcat):
        x, cat = ctx.saved_tensors
        dcat = dcat.contiguous()
        dx, dw, db = deconv2x2_bwd(x, ctx.wp_d, cat, dcat, 0, ctx.cout, False)
        if DEBUG_CAPTURE is not None:
            DEBUG_CAPTURE.append(dict(kind="deconv", x=x, w=ctx.wb_dbg[0], b=ctx.wb_dbg[1], cat=cat.clone(), dcat=dcat, dx=dx,
                                      dw=dw, db=db, coff=0, bf16=False))
        c0, c1 = ctx.cout, ctx.cout + ctx.ca
        return dx, dw, (db if ctx.has_b else None), dcat[..., c0:c1], dcat[..., c1:], None


class HeadLoss(_Op):
    """(xent, dice) = loss head over logits = z @ w + b; also returns logits / probs / metric sums
    as non-differentiable outputs."""

    @staticmethod
    def forward(ctx, z, w, b, labels, pixel_w, desc, want_probs):
        # undefined gradients stay None: autograd otherwise zero-fills a [N, H, W, classes] tensor per step for each of the
        # non-differentiable outputs (logits, probabilities: 2 x 25 MB at the headline shape) only to hand it to backward
        ctx.set_materialize_grads(False)
        w2 = w.reshape(desc.C, desc.ncls)
        logits, probs, result, ws = head_fwd(desc, z, w2, b, labels, pixel_w, want_probs)
        ctx.save_for_backward(z, w2, labels, pixel_w, logits, result, ws)
        ctx.desc = desc
        ctx.w_shape = w.shape
        ctx.sinks = (grad_sink(w, ctx), grad_sink(b, ctx))
        xent = result[0:1].reshape(())
        dice = result[1:2].reshape(())
        outs = (xent.clone(), dice.clone(), logits, probs if probs is not None else logits.new_empty(0), result)
        ctx.mark_non_differentiable(outs[2], outs[3], outs[4])
        return outs

    @staticmethod
    def backward(ctx, gx, gd, _gl, _gp, _gr):
        z, w2, labels, pixel_w, logits, result, ws = ctx.saved_tensors
        zero = torch.zeros((), dtype=torch.float32, device=z.device)
        scales = torch.stack([gx if gx is not None else zero, gd if gd is not None else zero]).to(torch.float32)
        sw, sb = _take(ctx.sinks[0]), _take(ctx.sinks[1])
        dz, dw, db = head_bwd(ctx.desc, z, w2, labels, pixel_w, logits, result, ws,
                              1.0 if gx is not None else 0.0, 1.0 if gd is not None else 0.0, scales,
                              out_w=sw.reshape(ctx.desc.C, ctx.desc.ncls) if sw is not None else None, out_b=sb)
        return dz, _ret(dw.reshape(ctx.w_shape), sw), _ret(db, sb), None, None, None, None
