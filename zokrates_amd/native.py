"""ctypes binding of libzkhip.so — the C ABI declared in include/zkhip.h.

This is plumbing: it adds no arithmetic.  If the HIP library is missing or no GPU is usable, every
entry point raises `ZkhipError`; there is no CPU path behind this module.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# ZKHIP_LIBRARY: development / test hook — load another build of the same C ABI (tests point bench.py at the TEST-ONLY
# emulator build to run its multi-rank plumbing on CPU); unset, the in-tree HIP library is the only candidate
DEFAULT_LIB = os.environ.get("ZKHIP_LIBRARY") or os.path.join(HERE, "libzkhip.so")

CURVE_IDS = {"bn128": 0, "bls12_381": 1}
FQ_BYTES = {0: 32, 1: 48}

ERR_NAMES = {0: "OK", -1: "BAD_ARG", -2: "PARSE", -3: "NOMEM", -4: "DEVICE", -5: "UNSATISFIED"}


class ZkhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"zkhip error {code} ({ERR_NAMES.get(code, '?')}): {msg}")
        self.code = code


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "h2d_ms", "matvec_ms", "ntt_ms", "msm_h_ms", "msm_z_ms", "finish_ms", "total_ms",
        "kernel_msm_accum_g1_ms", "kernel_msm_accum_g2_ms", "kernel_ntt_ms")] + [("reserved", C.c_float * 6)]

    def as_dict(self):
        return {n: float(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(buf, n=None):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf, dtype=np.uint8)
    if n is not None and a.size != n:
        raise ValueError(f"buffer has {a.size} bytes, expected {n}")
    return a


class Library:
    """One loaded copy of the C ABI."""

    SYMBOLS = [
        "zkhip_device_count", "zkhip_device_pci_bus_id", "zkhip_ctx_create", "zkhip_ctx_free", "zkhip_last_error", "zkhip_pk_load_g16",
        "zkhip_pk_free", "zkhip_pk_dims", "zkhip_r1cs_load", "zkhip_r1cs_free", "zkhip_prove_g16",
        "zkhip_prove_g16_batch", "zkhip_assignment_upload", "zkhip_assignment_free", "zkhip_prove_g16_resident", "zkhip_prove_g16_resident_batch",
        "zkhip_pk_load_g16_shard", "zkhip_partial_size", "zkhip_prove_g16_partial", "zkhip_combine_g16", "zkhip_ntt", "zkhip_witness_map", "zkhip_msm_g1", "zkhip_msm_g2",
        "zkhip_field_op", "zkhip_setup_g16_size", "zkhip_setup_g16", "zkhip_describe",
        "zkhip_pk_load_gm17", "zkhip_prove_gm17", "zkhip_prove_gm17_resident", "zkhip_prove_gm17_resident_batch",
        "zkhip_setup_gm17_size", "zkhip_setup_gm17", "zkhip_pk_load_gm17_shard", "zkhip_prove_gm17_partial", "zkhip_combine_gm17",
        "zkhip_prog_parse", "zkhip_prog_free", "zkhip_prog_dims", "zkhip_prog_matrix", "zkhip_prog_variable_order",
        "zkhip_prog_r1cs_load", "zkhip_prog_assignment", "zkhip_prog_write_bound", "zkhip_prog_write",
        "zkhip_pk_export_size", "zkhip_pk_export", "zkhip_pk_import",
        "zkhip_pk_bind_r1cs", "zkhip_pk_unbind", "zkhip_pk_is_bound", "zkhip_pk_bind_r1cs_shard", "zkhip_r1cs_fingerprint",
        "zkhip_ctx_tune", "zkhip_init", "zkhip_ctx_clock_probe", "zkhip_multi_bind", "zkhip_multi_unbind",
        "zkhip_multi_transform_split", "zkhip_multi_last_split", "zkhip_prove_g16_split_begin", "zkhip_prove_g16_split_end",
        "zkhip_ctx_create_multi", "zkhip_multi_free", "zkhip_multi_size", "zkhip_multi_ctx", "zkhip_multi_last_error", "zkhip_multi_r1cs_load",
        "zkhip_multi_pk_load_g16", "zkhip_multi_pk_load_gm17", "zkhip_prove_g16_multi", "zkhip_prove_gm17_multi",
        "zkhip_multi_pk_load_g16_replicas", "zkhip_prove_g16_multi_batch", "zkhip_multi_use_rccl", "zkhip_multi_exchange",
    ]

    def __init__(self, path=None):
        self.path = path or DEFAULT_LIB
        if not os.path.exists(self.path):
            raise ZkhipError(-4, f"{self.path} not found: build it with `python -m zokrates_amd.build` "
                                 "(libzkhip has no CPU fallback)")
        L = C.CDLL(self.path)
        vp, u64, u32, i32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_size_t
        pp = C.POINTER(C.c_void_p)
        L.zkhip_device_count.restype = i32
        L.zkhip_device_pci_bus_id.restype = i32; L.zkhip_device_pci_bus_id.argtypes = [i32, C.c_char_p, sz]
        L.zkhip_ctx_create.restype = i32; L.zkhip_ctx_create.argtypes = [i32, pp]
        L.zkhip_ctx_free.restype = None; L.zkhip_ctx_free.argtypes = [vp]
        L.zkhip_ctx_tune.restype = i32; L.zkhip_ctx_tune.argtypes = [vp, i32, i32]
        L.zkhip_ctx_create_multi.restype = i32; L.zkhip_ctx_create_multi.argtypes = [vp, i32, pp]
        L.zkhip_multi_free.restype = None; L.zkhip_multi_free.argtypes = [vp]
        L.zkhip_multi_size.restype = i32; L.zkhip_multi_size.argtypes = [vp]
        L.zkhip_multi_ctx.restype = vp; L.zkhip_multi_ctx.argtypes = [vp, i32]
        L.zkhip_multi_last_error.restype = C.c_char_p; L.zkhip_multi_last_error.argtypes = [vp]
        L.zkhip_multi_use_rccl.restype = i32; L.zkhip_multi_use_rccl.argtypes = [vp, i32]
        L.zkhip_multi_exchange.restype = C.c_char_p; L.zkhip_multi_exchange.argtypes = [vp]
        L.zkhip_multi_r1cs_load.restype = i32; L.zkhip_multi_r1cs_load.argtypes = [vp, i32, u64, u64, u64] + [vp] * 9
        L.zkhip_multi_pk_load_g16.restype = i32; L.zkhip_multi_pk_load_g16.argtypes = [vp, i32, vp, sz]
        L.zkhip_multi_pk_load_gm17.restype = i32; L.zkhip_multi_pk_load_gm17.argtypes = [vp, i32, vp, sz]
        L.zkhip_prove_g16_multi.restype = i32; L.zkhip_prove_g16_multi.argtypes = [vp] * 6
        L.zkhip_prove_gm17_multi.restype = i32; L.zkhip_prove_gm17_multi.argtypes = [vp] * 5
        L.zkhip_multi_pk_load_g16_replicas.restype = i32; L.zkhip_multi_pk_load_g16_replicas.argtypes = [vp, i32, vp, sz]
        L.zkhip_prove_g16_multi_batch.restype = i32; L.zkhip_prove_g16_multi_batch.argtypes = [vp, u32, vp, vp, vp, vp]
        L.zkhip_last_error.restype = C.c_char_p; L.zkhip_last_error.argtypes = [vp]
        L.zkhip_pk_load_g16.restype = i32; L.zkhip_pk_load_g16.argtypes = [vp, i32, vp, sz, pp]
        L.zkhip_pk_free.restype = None; L.zkhip_pk_free.argtypes = [vp]
        L.zkhip_pk_dims.restype = i32; L.zkhip_pk_dims.argtypes = [vp, vp]
        L.zkhip_r1cs_load.restype = i32; L.zkhip_r1cs_load.argtypes = [vp, i32, u64, u64, u64] + [vp] * 9 + [pp]
        L.zkhip_r1cs_free.restype = None; L.zkhip_r1cs_free.argtypes = [vp]
        L.zkhip_prove_g16.restype = i32; L.zkhip_prove_g16.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        L.zkhip_prove_g16_batch.restype = i32; L.zkhip_prove_g16_batch.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp]
        L.zkhip_assignment_upload.restype = i32; L.zkhip_assignment_upload.argtypes = [vp, vp, vp, pp]
        L.zkhip_assignment_free.restype = None; L.zkhip_assignment_free.argtypes = [vp]
        L.zkhip_prove_g16_resident.restype = i32; L.zkhip_prove_g16_resident.argtypes = [vp] * 8
        L.zkhip_prove_g16_resident_batch.restype = i32; L.zkhip_prove_g16_resident_batch.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp]
        L.zkhip_pk_load_g16_shard.restype = i32; L.zkhip_pk_load_g16_shard.argtypes = [vp, i32, vp, sz, u32, u32, pp]
        L.zkhip_partial_size.restype = i32; L.zkhip_partial_size.argtypes = [i32, vp]
        L.zkhip_prove_g16_partial.restype = i32; L.zkhip_prove_g16_partial.argtypes = [vp] * 9
        L.zkhip_combine_g16.restype = i32; L.zkhip_combine_g16.argtypes = [vp, vp, u32, vp, vp, vp, vp]
        L.zkhip_ntt.restype = i32; L.zkhip_ntt.argtypes = [vp, i32, u32, i32, vp]
        L.zkhip_witness_map.restype = i32; L.zkhip_witness_map.argtypes = [vp, vp, vp, vp]
        L.zkhip_msm_g1.restype = i32; L.zkhip_msm_g1.argtypes = [vp, i32, u64, vp, vp, vp]
        L.zkhip_msm_g2.restype = i32; L.zkhip_msm_g2.argtypes = [vp, i32, u64, vp, vp, vp]
        L.zkhip_field_op.restype = i32; L.zkhip_field_op.argtypes = [vp, i32, i32, i32, u64, vp, vp, vp]
        L.zkhip_setup_g16_size.restype = i32; L.zkhip_setup_g16_size.argtypes = [vp, vp]
        L.zkhip_setup_g16.restype = i32; L.zkhip_setup_g16.argtypes = [vp, vp, vp, vp, vp, vp, u64]
        L.zkhip_describe.restype = i32; L.zkhip_describe.argtypes = [vp, vp, sz]
        L.zkhip_pk_load_gm17.restype = i32; L.zkhip_pk_load_gm17.argtypes = [vp, i32, vp, sz, pp]
        L.zkhip_prove_gm17.restype = i32; L.zkhip_prove_gm17.argtypes = [vp] * 7
        L.zkhip_prove_gm17_resident.restype = i32; L.zkhip_prove_gm17_resident.argtypes = [vp] * 7
        L.zkhip_prove_gm17_resident_batch.restype = i32; L.zkhip_prove_gm17_resident_batch.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp]
        L.zkhip_pk_load_gm17_shard.restype = i32; L.zkhip_pk_load_gm17_shard.argtypes = [vp, i32, vp, sz, u32, u32, pp]
        L.zkhip_prove_gm17_partial.restype = i32; L.zkhip_prove_gm17_partial.argtypes = [vp] * 8
        L.zkhip_combine_gm17.restype = i32; L.zkhip_combine_gm17.argtypes = [vp, vp, u32, vp, vp, vp]
        L.zkhip_setup_gm17_size.restype = i32; L.zkhip_setup_gm17_size.argtypes = [vp, vp]
        L.zkhip_setup_gm17.restype = i32; L.zkhip_setup_gm17.argtypes = [vp, vp, vp, vp, vp, vp, u64]
        L.zkhip_pk_export_size.restype = i32; L.zkhip_pk_export_size.argtypes = [vp, vp]
        L.zkhip_pk_export.restype = i32; L.zkhip_pk_export.argtypes = [vp, vp, u64]
        L.zkhip_pk_import.restype = i32; L.zkhip_pk_import.argtypes = [vp, vp, sz, pp]
        L.zkhip_pk_bind_r1cs.restype = i32; L.zkhip_pk_bind_r1cs.argtypes = [vp, vp, vp]
        L.zkhip_pk_unbind.restype = i32; L.zkhip_pk_unbind.argtypes = [vp]
        L.zkhip_pk_is_bound.restype = i32; L.zkhip_pk_is_bound.argtypes = [vp, vp]
        L.zkhip_pk_bind_r1cs_shard.restype = i32; L.zkhip_pk_bind_r1cs_shard.argtypes = [vp, vp, vp, vp, sz]
        L.zkhip_r1cs_fingerprint.restype = i32; L.zkhip_r1cs_fingerprint.argtypes = [vp, vp, vp]
        L.zkhip_init.restype = i32; L.zkhip_init.argtypes = [i32]
        L.zkhip_ctx_clock_probe.restype = i32; L.zkhip_ctx_clock_probe.argtypes = [vp, u32, vp]
        L.zkhip_multi_bind.restype = i32; L.zkhip_multi_bind.argtypes = [vp, vp, sz]
        L.zkhip_multi_unbind.restype = i32; L.zkhip_multi_unbind.argtypes = [vp]
        L.zkhip_multi_transform_split.restype = i32; L.zkhip_multi_transform_split.argtypes = [vp, i32]
        L.zkhip_multi_last_split.restype = i32; L.zkhip_multi_last_split.argtypes = [vp]
        L.zkhip_prove_g16_split_begin.restype = i32; L.zkhip_prove_g16_split_begin.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp]
        L.zkhip_prove_g16_split_end.restype = i32; L.zkhip_prove_g16_split_end.argtypes = [vp, vp, vp, vp, vp, vp]
        L.zkhip_prog_parse.restype = i32; L.zkhip_prog_parse.argtypes = [vp, sz, pp]
        L.zkhip_prog_free.restype = None; L.zkhip_prog_free.argtypes = [vp]
        L.zkhip_prog_dims.restype = i32; L.zkhip_prog_dims.argtypes = [vp, vp]
        L.zkhip_prog_matrix.restype = i32; L.zkhip_prog_matrix.argtypes = [vp, i32, pp, pp, pp]
        L.zkhip_prog_variable_order.restype = i32; L.zkhip_prog_variable_order.argtypes = [vp, pp]
        L.zkhip_prog_r1cs_load.restype = i32; L.zkhip_prog_r1cs_load.argtypes = [vp, vp, pp]
        L.zkhip_prog_write_bound.restype = i32; L.zkhip_prog_write_bound.argtypes = [u64, u64, u64, vp]
        L.zkhip_prog_write.restype = i32; L.zkhip_prog_write.argtypes = [i32, u64, u64] + [vp] * 9 + [vp, vp, vp, u64, u32, vp, u64, vp]
        L.zkhip_prog_assignment.restype = i32; L.zkhip_prog_assignment.argtypes = [vp, vp, sz, vp, vp, u64, vp]
        self.L = L

    def init(self, hw_queues):
        """`zkhip_init`: the host's explicit request for `hw_queues` HIP hardware queues (GPU_MAX_HW_QUEUES, unless the process set it),
        BEFORE the first HIP call of the process.  The library never touches the environment by itself."""
        rc = self.L.zkhip_init(int(hw_queues))
        if rc != 0:
            raise ZkhipError(rc, self.L.zkhip_last_error(None).decode(errors="replace"))

    def device_count(self):
        return int(self.L.zkhip_device_count())

    def device_pci_bus_id(self, device):
        """"dddd:bb:dd.f" of HIP device `device`: /sys/bus/pci/devices/<that>/numa_node, .../hwmon (None if the runtime cannot tell)."""
        buf = C.create_string_buffer(32)
        if self.L.zkhip_device_pci_bus_id(int(device), buf, 32) != 0:
            return None
        return buf.value.decode().lower() or None


_default = None


def default_library():
    global _default
    if _default is None:
        _default = Library()
    return _default


class Context:
    """`zkhip_ctx`: one GPU, not re-entrant."""

    def __init__(self, device=0, library=None):
        self.lib = library or default_library()
        self.h = C.c_void_p()
        rc = self.lib.L.zkhip_ctx_create(device, C.byref(self.h))
        if rc != 0:
            raise ZkhipError(rc, self.lib.L.zkhip_last_error(None).decode())

    def _check(self, rc):
        if rc != 0:
            raise ZkhipError(rc, self.lib.L.zkhip_last_error(self.h).decode())

    def describe(self):
        buf = C.create_string_buffer(256)
        self._check(self.lib.L.zkhip_describe(self.h, buf, 256))
        return buf.value.decode()

    TUNABLES = {"msm_c": 1, "msm_waves": 2, "msm_lanes": 3, "msm_min_slice": 4, "fold_scan": 5, "serial": 6, "ntt_single_max_log": 7, "ntt_cols": 8, "slots": 9, "z_gate": 10, "fuse_z": 11, "msm_fused_waves": 12, "stream_jitter": 13, "ntt_max_sublog": 16, "msm_sets": 17, "skip_inf": 18, "b_sort": 19, "heavy_runs": 20, "lone_sched": 21, "ntt_skew_us": 22, "sort_two_level": 23, "fold_lines": 24, "fold_hg": 25, "ntt_fuse_first": 26, "fold_hop": 27, "pipe_plan": 28}

    def clock_probe(self, duration_us=2000):
        """`zkhip_ctx_clock_probe`: the shader clock (GHz) the device runs at over the next `duration_us` microseconds — one wavefront
        comparing the cycle counter with the wall clock beside whatever else the device is doing (call it on a SECOND context from a
        second thread while the first proves).  Blocks for the duration."""
        out = C.c_double()
        self._check(self.lib.L.zkhip_ctx_clock_probe(self.h, int(duration_us), C.byref(out)))
        return out.value

    def tune(self, name, value):
        """`zkhip_ctx_tune`: development / measurement knobs (window width, slices, fold form, serial streams ...)."""
        self._check(self.lib.L.zkhip_ctx_tune(self.h, self.TUNABLES[name], int(value)))

    def close(self):
        if self.h:
            self.lib.L.zkhip_ctx_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- primitives ----
    def ntt(self, curve_id, data, direction):
        d = np.array(_u8(data), dtype=np.uint8, copy=True)
        n = d.size // 32
        logn = n.bit_length() - 1
        if n == 0 or (1 << logn) != n:
            raise ValueError("NTT size must be a power of two")
        code = {"fft": 0, "ifft": 1, "coset_fft": 2, "coset_ifft": 3}[direction]
        self._check(self.lib.L.zkhip_ntt(self.h, curve_id, logn, code, _ptr(d)))
        return d

    def msm(self, curve_id, group, bases, scalars):
        nb = FQ_BYTES[curve_id]
        pt = 2 * nb * group
        scalars = _u8(scalars)
        n = scalars.size // 32
        bases = _u8(bases, n * pt)
        out = np.zeros(pt + 1, dtype=np.uint8)
        fn = self.lib.L.zkhip_msm_g1 if group == 1 else self.lib.L.zkhip_msm_g2
        self._check(fn(self.h, curve_id, n, _ptr(bases), _ptr(scalars), _ptr(out)))
        return out.tobytes()

    def field_op(self, curve_id, field, op, a, b):
        nb = 32 if field == 0 else FQ_BYTES[curve_id]
        a = _u8(a); b = _u8(b, a.size)
        out = np.zeros(a.size, dtype=np.uint8)
        code = {"add": 0, "sub": 1, "mul": 2}[op]
        self._check(self.lib.L.zkhip_field_op(self.h, curve_id, field, code, a.size // nb, _ptr(a), _ptr(b), _ptr(out)))
        return out


class ProvingKey:
    """`zkhip_pk`: an ark `proving.key` resident on the GPU in MSM-ready layout."""

    def __init__(self, ctx, curve_id, data, rank=0, world=1, scheme="g16"):
        """world > 1: load only rank's share of the bases (one proof sharded over several GPUs).
        scheme = "gm17": an ark-gm17 proving key (zkhip_pk_load_gm17; m = SAP variables, hlen = len(g_gamma2_z_t))."""
        self.ctx = ctx
        self.curve_id = curve_id
        self.rank, self.world = rank, world
        self.scheme = scheme
        data = _u8(data)
        self.h = C.c_void_p()
        if scheme == "gm17" and world == 1:
            ctx._check(ctx.lib.L.zkhip_pk_load_gm17(ctx.h, curve_id, _ptr(data), data.size, C.byref(self.h)))
        elif scheme == "gm17":
            ctx._check(ctx.lib.L.zkhip_pk_load_gm17_shard(ctx.h, curve_id, _ptr(data), data.size, rank, world, C.byref(self.h)))
        elif world == 1:
            ctx._check(ctx.lib.L.zkhip_pk_load_g16(ctx.h, curve_id, _ptr(data), data.size, C.byref(self.h)))
        else:
            ctx._check(ctx.lib.L.zkhip_pk_load_g16_shard(ctx.h, curve_id, _ptr(data), data.size, rank, world, C.byref(self.h)))
        self._dims()

    def _dims(self):
        d = np.zeros(4, dtype=np.uint64)
        self.ctx._check(self.ctx.lib.L.zkhip_pk_dims(self.h, _ptr(d)))
        self.m, self.hlen, self.w, self.l = (int(x) for x in d)

    def export_image(self):
        """The resident (device-layout) form of this key as bytes (`zkhip_pk_export`): what a key cache stores — level 0 of the
        five base tables; the import recomputes the window multiples on the device."""
        size = C.c_uint64()
        self.ctx._check(self.ctx.lib.L.zkhip_pk_export_size(self.h, C.byref(size)))
        out = np.zeros(size.value, dtype=np.uint8)
        self.ctx._check(self.ctx.lib.L.zkhip_pk_export(self.h, _ptr(out), size.value))
        return out

    @classmethod
    def from_image(cls, ctx, curve_id, image, scheme="g16"):
        """`zkhip_pk_import`: no parsing, no conversion — five host-to-device copies."""
        self = cls.__new__(cls)
        self.ctx, self.curve_id, self.rank, self.world, self.scheme = ctx, curve_id, 0, 1, scheme
        image = _u8(image)
        self.h = C.c_void_p()
        ctx._check(ctx.lib.L.zkhip_pk_import(ctx.h, _ptr(image), image.size, C.byref(self.h)))
        self._dims()
        return self

    def bind(self, cs):
        """`zkhip_pk_bind_r1cs`: apply the quotient's transforms to this key's bases once, for the constraint system `cs` — proofs
        over (this key, cs) then take four transforms instead of six and skip c; the proof bytes do not change."""
        self.ctx._check(self.ctx.lib.L.zkhip_pk_bind_r1cs(self.ctx.h, self.h, cs.h))

    def bind_shard(self, cs, key_bytes):
        """`zkhip_pk_bind_r1cs_shard`: the same for a SHARD (or a whole key) from the key file — the transforms need every base of
        the key once; this key keeps its index ranges of the bound tables."""
        data = _u8(key_bytes)
        self.ctx._check(self.ctx.lib.L.zkhip_pk_bind_r1cs_shard(self.ctx.h, self.h, cs.h, _ptr(data), data.size))

    def unbind(self):
        self.ctx._check(self.ctx.lib.L.zkhip_pk_unbind(self.h))

    def is_bound(self, cs):
        return bool(self.ctx.lib.L.zkhip_pk_is_bound(self.h, cs.h))

    def close(self):
        if self.h:
            self.ctx.lib.L.zkhip_pk_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ConstraintSystem:
    """`zkhip_r1cs`: CSR matrices in ark variable order, resident on the GPU."""

    def __init__(self, ctx, curve_id, n, l, w, mats):
        self.ctx = ctx
        self.curve_id = curve_id
        self.n, self.l, self.w = n, l, w
        self.m = l + w
        args, keep = [], []
        for rp, col, val in mats:
            rp = np.ascontiguousarray(rp, dtype=np.uint64)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            val = _u8(val)
            if rp.size != n + 1 or val.size != col.size * 32:
                raise ValueError("CSR shape mismatch")
            keep += [rp, col, val]
            args += [_ptr(rp), _ptr(col), _ptr(val)]
        self.h = C.c_void_p()
        ctx._check(ctx.lib.L.zkhip_r1cs_load(ctx.h, curve_id, n, l, w, *args, C.byref(self.h)))

    def fingerprint(self):
        """`zkhip_r1cs_fingerprint`: the checksum a key image remembers of the system its bound tables were made for."""
        out = np.zeros(2, dtype=np.uint64)
        self.ctx._check(self.ctx.lib.L.zkhip_r1cs_fingerprint(self.ctx.h, self.h, _ptr(out)))
        return int(out[0]), int(out[1])

    def witness_map(self, z):
        N = 1
        while N < self.n + self.l:
            N *= 2
        z = _u8(z, self.m * 32)
        out = np.zeros(N * 32, dtype=np.uint8)
        self.ctx._check(self.ctx.lib.L.zkhip_witness_map(self.ctx.h, self.h, _ptr(z), _ptr(out)))
        return out

    def close(self):
        if self.h:
            self.ctx.lib.L.zkhip_r1cs_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Assignment:
    """`zkhip_assignment`: a full assignment z (m x 32 B canonical LE, z[0] = 1) resident in HBM."""

    def __init__(self, ctx, cs, z):
        self.ctx = ctx
        z = _u8(z, cs.m * 32)
        self.h = C.c_void_p()
        ctx._check(ctx.lib.L.zkhip_assignment_upload(ctx.h, cs.h, _ptr(z), C.byref(self.h)))

    def close(self):
        if self.h:
            self.ctx.lib.L.zkhip_assignment_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prove_g16_resident(ctx, pk, cs, assignment, r, s, want_timings=False):
    """Like prove_g16 with the assignment already on the GPU."""
    nb = FQ_BYTES[pk.curve_id]
    rb = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8)
    sb = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8)
    out = np.zeros(8 * nb + 3, dtype=np.uint8)
    tm = Timings()
    ctx._check(ctx.lib.L.zkhip_prove_g16_resident(ctx.h, pk.h, cs.h, assignment.h, _ptr(rb), _ptr(sb), _ptr(out), C.byref(tm)))
    return (out.tobytes(), tm.as_dict()) if want_timings else out.tobytes()


def prove_g16_split_begin(ctx, pk_shard, cs, z, r, s, half):
    """`zkhip_prove_g16_split_begin`: this rank's half (0: a, 1: b on the coset) of a split witness map, as uint8[N * 32]; the proof
    stays in flight until prove_g16_split_end."""
    n_dom = pk_shard.hlen + 1
    out = np.zeros(n_dom * 32, dtype=np.uint8)
    rb, sb = int(r).to_bytes(32, "little"), int(s).to_bytes(32, "little")
    if isinstance(z, Assignment):
        zp, za = None, z.h
    else:
        zp, za = _ptr(_u8(z, cs.m * 32)), None
    ctx._check(ctx.lib.L.zkhip_prove_g16_split_begin(ctx.h, pk_shard.h, cs.h, zp, za, rb, sb, int(half), _ptr(out)))
    return out


def prove_g16_split_end(ctx, pk_shard, cs, other_half):
    """`zkhip_prove_g16_split_end`: the partner's half in, this rank's partial record out."""
    other = _u8(other_half, (pk_shard.hlen + 1) * 32)
    out = np.zeros(partial_size(ctx, pk_shard.curve_id), dtype=np.uint8)
    tm = Timings()
    ctx._check(ctx.lib.L.zkhip_prove_g16_split_end(ctx.h, pk_shard.h, cs.h, _ptr(other), _ptr(out), C.byref(tm)))
    return out


def partial_size(ctx, curve_id):
    n = C.c_uint64()
    ctx._check(ctx.lib.L.zkhip_partial_size(curve_id, C.byref(n)))
    return int(n.value)


def prove_g16_partial(ctx, pk_shard, cs, z, r, s, want_timings=False):
    """One rank's share of a proof: z is a host assignment (uint8[m*32]) or an `Assignment`.  Returns the partial record."""
    rb = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8)
    sb = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8)
    out = np.zeros(partial_size(ctx, pk_shard.curve_id), dtype=np.uint8)
    tm = Timings()
    if isinstance(z, Assignment):
        zp, za = None, z.h
    else:
        z = _u8(z, cs.m * 32)
        zp, za = _ptr(z), None
    ctx._check(ctx.lib.L.zkhip_prove_g16_partial(ctx.h, pk_shard.h, cs.h, zp, za, _ptr(rb), _ptr(sb), _ptr(out), C.byref(tm)))
    return (out, tm.as_dict()) if want_timings else out


def combine_g16(ctx, pk, partials, r, s):
    """Adds the ranks' partial records (list of uint8 arrays) and assembles the proof."""
    nb = FQ_BYTES[pk.curve_id]
    buf = np.ascontiguousarray(np.concatenate([_u8(p) for p in partials]))
    rb = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8)
    sb = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8)
    out = np.zeros(8 * nb + 3, dtype=np.uint8)
    ctx._check(ctx.lib.L.zkhip_combine_g16(ctx.h, pk.h, len(partials), _ptr(buf), _ptr(rb), _ptr(sb), _ptr(out)))
    return out.tobytes()


def setup_g16(ctx, cs, toxic, g1=None, g2=None):
    """Groth16 setup on the GPU: `toxic` = (alpha, beta, gamma, delta, tau) ints; returns the ark-format proving key bytes."""
    size = C.c_uint64()
    ctx._check(ctx.lib.L.zkhip_setup_g16_size(cs.h, C.byref(size)))
    tb = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in toxic), dtype=np.uint8)
    out = np.zeros(size.value, dtype=np.uint8)
    g1p = _ptr(_u8(g1)) if g1 is not None else None
    g2p = _ptr(_u8(g2)) if g2 is not None else None
    ctx._check(ctx.lib.L.zkhip_setup_g16(ctx.h, cs.h, _ptr(tb), g1p, g2p, _ptr(out), size.value))
    return out


def prove_g16(ctx, pk, cs, z, r, s, want_timings=False):
    """Raw proof bytes (8*sz(Fq)+3) for assignment z (m*32 B canonical LE) and blinding scalars r, s (ints)."""
    nb = FQ_BYTES[pk.curve_id]
    z = _u8(z, cs.m * 32)
    rb = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8)
    sb = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8)
    out = np.zeros(8 * nb + 3, dtype=np.uint8)
    tm = Timings()
    ctx._check(ctx.lib.L.zkhip_prove_g16(ctx.h, pk.h, cs.h, _ptr(z), _ptr(rb), _ptr(sb), _ptr(out), C.byref(tm)))
    return (out.tobytes(), tm.as_dict()) if want_timings else out.tobytes()


def prove_g16_resident_batch(ctx, pk, cs, assignments, rs):
    """Pipelined proofs over resident assignments (may repeat); rs: list of (r, s) ints.  Returns (proofs, timings)."""
    nb = FQ_BYTES[pk.curve_id]
    count = len(rs)
    assert len(assignments) == count
    handles = (C.c_void_p * count)(*[a.h for a in assignments])
    rsb = np.frombuffer(b"".join(int(r).to_bytes(32, "little") + int(s).to_bytes(32, "little") for r, s in rs), dtype=np.uint8)
    out = np.zeros(count * (8 * nb + 3), dtype=np.uint8)
    tm = Timings()
    ctx._check(ctx.lib.L.zkhip_prove_g16_resident_batch(ctx.h, pk.h, cs.h, count, handles, _ptr(rsb), _ptr(out), C.byref(tm)))
    step = 8 * nb + 3
    return [out[i * step:(i + 1) * step].tobytes() for i in range(count)], tm.as_dict()


def prove_g16_batch(ctx, pk, cs, zs, rs):
    """zs: uint8[count*m*32]; rs: list of (r, s) ints.  Returns (list of raw proofs, summed timings)."""
    nb = FQ_BYTES[pk.curve_id]
    count = len(rs)
    zs = _u8(zs, count * pk.m * 32)
    rsb = np.frombuffer(b"".join(int(r).to_bytes(32, "little") + int(s).to_bytes(32, "little") for r, s in rs), dtype=np.uint8)
    out = np.zeros(count * (8 * nb + 3), dtype=np.uint8)
    tm = Timings()
    ctx._check(ctx.lib.L.zkhip_prove_g16_batch(ctx.h, pk.h, cs.h, count, _ptr(zs), _ptr(rsb), _ptr(out), C.byref(tm)))
    step = 8 * nb + 3
    return [out[i * step:(i + 1) * step].tobytes() for i in range(count)], tm.as_dict()


# ---- GM17 (BASELINE.json config 5) ----
def _rnd96(d1, d2, r):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in (d1, d2, r)), dtype=np.uint8)


def setup_gm17(ctx, cs, toxic, g1=None, g2=None):
    """GM17 setup on the GPU: `toxic` = (alpha, beta, gamma, t) ints; returns the ark-gm17-format proving key bytes."""
    size = C.c_uint64()
    ctx._check(ctx.lib.L.zkhip_setup_gm17_size(cs.h, C.byref(size)))
    tb = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in toxic), dtype=np.uint8)
    out = np.zeros(size.value, dtype=np.uint8)
    g1p = _ptr(_u8(g1)) if g1 is not None else None
    g2p = _ptr(_u8(g2)) if g2 is not None else None
    ctx._check(ctx.lib.L.zkhip_setup_gm17(ctx.h, cs.h, _ptr(tb), g1p, g2p, _ptr(out), size.value))
    return out


def prove_gm17(ctx, pk, cs, z, d1, d2, r, want_timings=False):
    """Raw proof bytes (8*sz(Fq)+3) for assignment z (host bytes or a resident `Assignment`) and blinding scalars d1, d2, r."""
    nb = FQ_BYTES[pk.curve_id]
    rnd = _rnd96(d1, d2, r)
    out = np.zeros(8 * nb + 3, dtype=np.uint8)
    tm = Timings()
    if isinstance(z, Assignment):
        ctx._check(ctx.lib.L.zkhip_prove_gm17_resident(ctx.h, pk.h, cs.h, z.h, _ptr(rnd), _ptr(out), C.byref(tm)))
    else:
        z = _u8(z, cs.m * 32)
        ctx._check(ctx.lib.L.zkhip_prove_gm17(ctx.h, pk.h, cs.h, _ptr(z), _ptr(rnd), _ptr(out), C.byref(tm)))
    return (out.tobytes(), tm.as_dict()) if want_timings else out.tobytes()


def prove_gm17_partial(ctx, pk_shard, cs, z, d1, d2, r):
    """One rank's share of a GM17 proof: z is a host assignment or an `Assignment`.  Returns the partial record."""
    rnd = _rnd96(d1, d2, r)
    out = np.zeros(partial_size(ctx, pk_shard.curve_id), dtype=np.uint8)
    tm = Timings()
    if isinstance(z, Assignment):
        zp, za = None, z.h
    else:
        z = _u8(z, cs.m * 32)
        zp, za = _ptr(z), None
    ctx._check(ctx.lib.L.zkhip_prove_gm17_partial(ctx.h, pk_shard.h, cs.h, zp, za, _ptr(rnd), _ptr(out), C.byref(tm)))
    return out


def combine_gm17(ctx, pk, partials, d1, d2, r):
    nb = FQ_BYTES[pk.curve_id]
    buf = np.ascontiguousarray(np.concatenate([_u8(p) for p in partials]))
    rnd = _rnd96(d1, d2, r)
    out = np.zeros(8 * nb + 3, dtype=np.uint8)
    ctx._check(ctx.lib.L.zkhip_combine_gm17(ctx.h, pk.h, len(partials), _ptr(buf), _ptr(rnd), _ptr(out)))
    return out.tobytes()


def prove_gm17_resident_batch(ctx, pk, cs, assignments, rnds):
    """Pipelined GM17 proofs over resident assignments (may repeat); rnds: list of (d1, d2, r) ints."""
    nb = FQ_BYTES[pk.curve_id]
    count = len(rnds)
    assert len(assignments) == count
    handles = (C.c_void_p * count)(*[a.h for a in assignments])
    rb = np.ascontiguousarray(np.concatenate([_rnd96(*t) for t in rnds])) if count else np.zeros(1, dtype=np.uint8)
    out = np.zeros(max(count, 1) * (8 * nb + 3), dtype=np.uint8)
    tm = Timings()
    ctx._check(ctx.lib.L.zkhip_prove_gm17_resident_batch(ctx.h, pk.h, cs.h, count, handles, _ptr(rb), _ptr(out), C.byref(tm)))
    step = 8 * nb + 3
    return [out[i * step:(i + 1) * step].tobytes() for i in range(count)], tm.as_dict()


# ---- ZoKrates' own files (N1): host only ----
class Program:
    """`zkhip_prog`: a ZoKrates `out` file parsed into the R1CS in ark variable order (no GPU involved)."""

    def __init__(self, data, library=None):
        self.lib = library or default_library()
        data = _u8(data)
        self.h = C.c_void_p()
        rc = self.lib.L.zkhip_prog_parse(_ptr(data), data.size, C.byref(self.h))
        if rc != 0:
            raise ZkhipError(rc, self.lib.L.zkhip_last_error(None).decode())
        d = np.zeros(8, dtype=np.uint64)
        self.lib.L.zkhip_prog_dims(self.h, _ptr(d))
        self.curve_id, self.n, self.l, self.w, self.return_count, self.n_public_args, self.nnz = (int(x) for x in d[:7])
        self.m = self.l + self.w

    def mats(self):
        """[(rowptr u64[n+1], col u32[nnz], val u8[nnz*32])] for A, B, C — copies."""
        out = []
        for k in range(3):
            rp, col, val = C.c_void_p(), C.c_void_p(), C.c_void_p()
            self.lib.L.zkhip_prog_matrix(self.h, k, C.byref(rp), C.byref(col), C.byref(val))
            rpa = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_uint64)), (self.n + 1,)).copy()
            nnz = int(rpa[-1])
            if nnz:
                cola = np.ctypeslib.as_array(C.cast(col, C.POINTER(C.c_uint32)), (nnz,)).copy()
                vala = np.ctypeslib.as_array(C.cast(val, C.POINTER(C.c_uint8)), (nnz * 32,)).copy()
            else:
                cola, vala = np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint8)
            out.append((rpa, cola, vala))
        return out

    def variable_order(self):
        ids = C.c_void_p()
        self.lib.L.zkhip_prog_variable_order(self.h, C.byref(ids))
        return np.ctypeslib.as_array(C.cast(ids, C.POINTER(C.c_int64)), (self.m,)).copy()

    def constraint_system(self, ctx):
        """Upload to the GPU (`zkhip_prog_r1cs_load`)."""
        cs = ConstraintSystem.__new__(ConstraintSystem)
        cs.ctx, cs.curve_id, cs.n, cs.l, cs.w, cs.m = ctx, self.curve_id, self.n, self.l, self.w, self.m
        cs.h = C.c_void_p()
        ctx._check(ctx.lib.L.zkhip_prog_r1cs_load(ctx.h, self.h, C.byref(cs.h)))
        return cs

    def assignment(self, witness_bytes):
        """(z uint8[m*32] in ark order, inputs uint8[k*32] = public_inputs_values) from a ZoKrates `witness` file."""
        wit = _u8(witness_bytes)
        z = np.zeros(self.m * 32, dtype=np.uint8)
        cap = self.n_public_args + wit.size // 40 + 1
        inputs = np.zeros(cap * 32, dtype=np.uint8)
        n_in = C.c_uint64()
        rc = self.lib.L.zkhip_prog_assignment(self.h, _ptr(wit), wit.size, _ptr(z), _ptr(inputs), cap, C.byref(n_in))
        if rc != 0:
            raise ZkhipError(rc, self.lib.L.zkhip_last_error(None).decode())
        return z, inputs[:32 * n_in.value].copy()

    def close(self):
        if self.h:
            self.lib.L.zkhip_prog_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_program(curve_id, n, m, mats, ids=None, args=((1, False),), return_count=0, library=None):
    """`zkhip_prog_write`: the R1CS `mats` (as for ConstraintSystem) as the bytes of a ZoKrates `out` program.  ids[j] = the
    ZoKrates variable id of column j (default j: column 0 = ~one, column j = _{j-1}); args = [(id, private)]."""
    lib = library or default_library()
    ids = np.arange(m, dtype=np.int64) if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
    if ids.size != m:
        raise ValueError("one id per column")
    keep, a, nnz = [], [], 0
    for rp, col, val in mats:
        rp = np.ascontiguousarray(rp, dtype=np.uint64); col = np.ascontiguousarray(col, dtype=np.uint32); val = _u8(val)
        if rp.size != n + 1 or val.size != col.size * 32:
            raise ValueError("CSR shape mismatch")
        keep += [rp, col, val]
        a += [_ptr(rp), _ptr(col), _ptr(val)]
        nnz += int(col.size)
    arg_ids = np.asarray([i for i, _ in args], dtype=np.int64)
    arg_priv = np.asarray([1 if p else 0 for _, p in args], dtype=np.uint8)
    bound = C.c_uint64()
    lib.L.zkhip_prog_write_bound(n, nnz, len(args), C.byref(bound))
    out = np.empty(bound.value, dtype=np.uint8)
    ln = C.c_uint64()
    rc = lib.L.zkhip_prog_write(curve_id, n, m, *a, _ptr(ids), _ptr(arg_ids), _ptr(arg_priv), len(args), return_count, _ptr(out), bound.value, C.byref(ln))
    if rc != 0:
        raise ZkhipError(rc, lib.L.zkhip_last_error(None).decode())
    return out[:ln.value]


def write_witness(ids, z):
    """`Witness::write` (/root/reference/zokrates_ast/src/ir/witness.rs:44-53): usize count, then (isize id, 32-byte canonical
    LE value) in ascending signed-id order (the BTreeMap's).  ids[j] names entry j of z (uint8[m*32])."""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    z = _u8(z, ids.size * 32).reshape(-1, 32)
    order = np.argsort(ids, kind="stable")
    rec = np.zeros(ids.size, dtype=[("id", "<i8"), ("v", "u1", 32)])
    rec["id"] = ids[order]
    rec["v"] = z[order]
    return np.concatenate([np.frombuffer(int(ids.size).to_bytes(8, "little"), dtype=np.uint8), rec.view(np.uint8).reshape(-1)])


class Multi:
    """`zkhip_multi`: ONE proof across several GPUs of this process (no collective library, no Python in the data path):
    member k holds shard k of n of the key and a replica of the constraint system.  `devices` may repeat a device."""

    def __init__(self, devices, library=None):
        self.lib = library or default_library()
        self.h = C.c_void_p()
        dev = np.asarray(list(devices), dtype=np.int32)
        rc = self.lib.L.zkhip_ctx_create_multi(_ptr(dev), int(dev.size), C.byref(self.h))
        if rc != 0:
            raise ZkhipError(rc, self.lib.L.zkhip_last_error(None).decode())
        self.curve_id = None
        self.m = None

    def _check(self, rc):
        if rc != 0:
            raise ZkhipError(rc, self.lib.L.zkhip_multi_last_error(self.h).decode())

    def __len__(self):
        return int(self.lib.L.zkhip_multi_size(self.h))

    def member_context(self, k=0):
        """A borrowed `Context` of member k (do not close it): e.g. to run the setup on the same device."""
        ctx = Context.__new__(Context)
        ctx.lib = self.lib
        ctx.h = C.c_void_p(self.lib.L.zkhip_multi_ctx(self.h, k))
        ctx.close = lambda: None
        return ctx

    def use_rccl(self, on=True):
        """`zkhip_multi_use_rccl`: exchange the members' shares with an RCCL all-gather instead of through host memory."""
        self._check(self.lib.L.zkhip_multi_use_rccl(self.h, 1 if on else 0))

    def exchange(self):
        return self.lib.L.zkhip_multi_exchange(self.h).decode()

    def load_constraint_system(self, curve_id, n, l, w, mats):
        keep = []
        args = []
        for rp, col, val in mats:
            rp = np.ascontiguousarray(rp, dtype=np.uint64); col = np.ascontiguousarray(col, dtype=np.uint32); val = _u8(val)
            keep += [rp, col, val]
            args += [_ptr(rp), _ptr(col), _ptr(val)]
        self._check(self.lib.L.zkhip_multi_r1cs_load(self.h, curve_id, n, l, w, *args))
        self.curve_id, self.m = curve_id, l + w

    def load_proving_key(self, curve_id, pk_bytes, scheme="g16"):
        b = _u8(pk_bytes)
        fn = self.lib.L.zkhip_multi_pk_load_gm17 if scheme == "gm17" else self.lib.L.zkhip_multi_pk_load_g16
        self._check(fn(self.h, curve_id, _ptr(b), b.size))
        self.curve_id = curve_id

    def load_proving_key_replicas(self, curve_id, pk_bytes):
        """Throughput mode: the whole Groth16 key on every member (`zkhip_multi_pk_load_g16_replicas`)."""
        b = _u8(pk_bytes)
        self._check(self.lib.L.zkhip_multi_pk_load_g16_replicas(self.h, curve_id, _ptr(b), b.size))
        self.curve_id = curve_id

    def bind(self, pk_bytes):
        """`zkhip_multi_bind`: the members' keys bound to the members' constraint system (one member computes the bound bases from
        the key file, every member installs its index ranges)."""
        b = _u8(pk_bytes)
        self._check(self.lib.L.zkhip_multi_bind(self.h, _ptr(b), b.size))

    def unbind(self):
        self._check(self.lib.L.zkhip_multi_unbind(self.h))

    def transform_split(self, on=None):
        """`zkhip_multi_transform_split`: let bound members split a proof's witness map (None: only report).  Returns the previous setting."""
        return bool(self.lib.L.zkhip_multi_transform_split(self.h, -1 if on is None else int(bool(on))))

    def last_split(self):
        return bool(self.lib.L.zkhip_multi_last_split(self.h))

    def prove_g16_batch(self, zs, rss):
        """Independent proofs dealt over the members (`zkhip_prove_g16_multi_batch`): zs = list of host assignments,
        rss = list of (r, s).  Returns (list of proof bytes, timings dict)."""
        nb = FQ_BYTES[self.curve_id]
        count = len(zs)
        z = np.ascontiguousarray(np.concatenate([_u8(a, self.m * 32) for a in zs])) if count else np.zeros(1, dtype=np.uint8)
        rs = np.frombuffer(b"".join(int(r).to_bytes(32, "little") + int(s).to_bytes(32, "little") for r, s in rss) or b"\0", dtype=np.uint8)
        out = np.zeros(max(count, 1) * (8 * nb + 3), dtype=np.uint8)
        tm = Timings()
        self._check(self.lib.L.zkhip_prove_g16_multi_batch(self.h, count, _ptr(z), _ptr(rs), _ptr(out), C.byref(tm)))
        pb = 8 * nb + 3
        return [out[i * pb:(i + 1) * pb].tobytes() for i in range(count)], tm.as_dict()

    def prove_g16(self, z, r, s, want_timings=False):
        nb = FQ_BYTES[self.curve_id]
        z = _u8(z, self.m * 32)
        rb = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8)
        sb = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8)
        out = np.zeros(8 * nb + 3, dtype=np.uint8)
        tm = Timings()
        self._check(self.lib.L.zkhip_prove_g16_multi(self.h, _ptr(z), _ptr(rb), _ptr(sb), _ptr(out), C.byref(tm)))
        return (out.tobytes(), tm.as_dict()) if want_timings else out.tobytes()

    def prove_gm17(self, z, d1, d2, r, want_timings=False):
        nb = FQ_BYTES[self.curve_id]
        z = _u8(z, self.m * 32)
        rnd = _rnd96(d1, d2, r)
        out = np.zeros(8 * nb + 3, dtype=np.uint8)
        tm = Timings()
        self._check(self.lib.L.zkhip_prove_gm17_multi(self.h, _ptr(z), _ptr(rnd), _ptr(out), C.byref(tm)))
        return (out.tobytes(), tm.as_dict()) if want_timings else out.tobytes()

    def close(self):
        if self.h:
            self.lib.L.zkhip_multi_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
