"""ctypes binding of oracle/c/liboracle_g16.so (the fast CPU restatement / CPU baseline).
TEST ORACLE ONLY — see oracle/__init__.py for who may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "c", "liboracle_g16.so")
_lib = None


def _cpu_tag():
    """The library is compiled with -march=native: rebuild when the host CPU differs from the build host
    (the .so travels to the GPU box with the repository snapshot)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    import hashlib
                    return hashlib.sha1(line.encode()).hexdigest()
    except OSError:
        pass
    return "unknown"


def build(force=False):
    srcs = [os.path.join(HERE, "c", f) for f in ("oracle_g16.cpp", "ff.hpp", "ec.hpp", "gm17.hpp")]
    stamp = LIB_PATH + ".cpu"
    tag = _cpu_tag()
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if (force or have != tag or not os.path.exists(LIB_PATH)
            or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B"])
        with open(stamp, "w") as f:
            f.write(tag)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, u64, i32, u8p = C.c_void_p, C.c_uint64, C.c_int, C.c_char_p
        L.orc_circuit_synth.restype = vp; L.orc_circuit_synth.argtypes = [i32, u64, u64, i32]
        L.orc_circuit_from_csr.restype = vp
        L.orc_circuit_from_csr.argtypes = [i32, u64, u64, u64] + [vp] * 9
        L.orc_circuit_free.argtypes = [vp]
        L.orc_circuit_dims.argtypes = [vp, vp]
        L.orc_circuit_export.argtypes = [vp, i32, vp, vp, vp]
        L.orc_circuit_assignment.argtypes = [vp, vp]
        L.orc_setup.restype = vp; L.orc_setup.argtypes = [vp, u8p, i32]
        L.orc_pk_parse.restype = vp; L.orc_pk_parse.argtypes = [i32, u8p, u64]
        L.orc_pk_size.restype = u64; L.orc_pk_size.argtypes = [vp]
        L.orc_pk_serialize.argtypes = [vp, vp]
        L.orc_pk_free.argtypes = [vp]
        L.orc_prove.restype = i32; L.orc_prove.argtypes = [vp, vp, vp, u8p, u8p, vp, i32, vp]
        L.orc_trapdoor.restype = i32; L.orc_trapdoor.argtypes = [vp, u8p, vp, u8p, u8p, vp, i32]
        L.orc_witness_map.restype = i32; L.orc_witness_map.argtypes = [vp, vp, vp, i32]
        L.orc_ntt.restype = i32; L.orc_ntt.argtypes = [i32, i32, i32, vp, i32]
        L.orc_msm.restype = i32; L.orc_msm.argtypes = [i32, i32, u64, vp, vp, vp, i32]
        L.orc_field_op.restype = i32; L.orc_field_op.argtypes = [i32, i32, i32, u8p, u8p, vp]
        L.orc_gm17_setup.restype = vp; L.orc_gm17_setup.argtypes = [vp, u8p, i32]
        L.orc_gm17_pk_parse.restype = vp; L.orc_gm17_pk_parse.argtypes = [i32, vp, u64]
        L.orc_gm17_pk_size.restype = u64; L.orc_gm17_pk_size.argtypes = [vp]
        L.orc_gm17_pk_serialize.argtypes = [vp, vp]
        L.orc_gm17_prove.restype = i32; L.orc_gm17_prove.argtypes = [vp, vp, vp, u8p, u8p, u8p, vp, i32, vp]
        L.orc_gm17_trapdoor.restype = i32; L.orc_gm17_trapdoor.argtypes = [vp, u8p, vp, u8p, u8p, vp, i32]
        L.orc_generators.argtypes = [i32, vp, vp]
        L.orc_hardware_threads.restype = i32
        _lib = L
    return _lib


def hw_threads():
    return lib().orc_hardware_threads()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


FQ_BYTES = {0: 32, 1: 48}


class Circuit:
    """R1CS in ark variable order + (for synthetic circuits) a satisfying assignment."""

    def __init__(self, handle, curve_id):
        self.h = handle
        self.curve_id = curve_id
        d = np.zeros(7, dtype=np.uint64)
        lib().orc_circuit_dims(self.h, _ptr(d))
        self.n, self.l, self.w, self.nnzA, self.nnzB, self.nnzC, self.N = (int(x) for x in d)
        self.m = self.l + self.w

    @staticmethod
    def synth(curve_id, n, seed, kind="dense"):
        return Circuit(lib().orc_circuit_synth(curve_id, n, seed, 1 if kind == "sha" else 0), curve_id)

    @staticmethod
    def from_csr(curve_id, n, l, w, mats):
        """mats = [(rowptr u64[n+1], col u32[nnz], val u8[nnz*32])] * 3"""
        args = []
        keep = []
        for rp, col, val in mats:
            rp = np.ascontiguousarray(rp, dtype=np.uint64)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            val = np.ascontiguousarray(val, dtype=np.uint8)
            keep += [rp, col, val]
            args += [_ptr(rp), _ptr(col), _ptr(val)]
        return Circuit(lib().orc_circuit_from_csr(curve_id, n, l, w, *args), curve_id)

    def csr(self, which):
        nnz = (self.nnzA, self.nnzB, self.nnzC)[which]
        rp = np.zeros(self.n + 1, dtype=np.uint64)
        col = np.zeros(max(nnz, 1), dtype=np.uint32)
        val = np.zeros(max(nnz, 1) * 32, dtype=np.uint8)
        lib().orc_circuit_export(self.h, which, _ptr(rp), _ptr(col), _ptr(val))
        return rp, col[:nnz], val[:nnz * 32]

    def assignment(self):
        z = np.zeros(self.m * 32, dtype=np.uint8)
        lib().orc_circuit_assignment(self.h, _ptr(z))
        return z

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_circuit_free(self.h)
            self.h = None


def toxic_bytes(tox):
    return b"".join(int(v).to_bytes(32, "little") for v in (tox.alpha, tox.beta, tox.gamma, tox.delta, tox.tau))


class ProvingKey:
    def __init__(self, handle, curve_id):
        self.h = handle
        self.curve_id = curve_id

    @staticmethod
    def setup(circuit, toxic, threads=0):
        return ProvingKey(lib().orc_setup(circuit.h, toxic, threads or hw_threads()), circuit.curve_id)

    @staticmethod
    def parse(curve_id, data):
        h = lib().orc_pk_parse(curve_id, bytes(data), len(data))
        if not h:
            raise ValueError("malformed proving key")
        return ProvingKey(h, curve_id)

    def serialize(self):
        n = lib().orc_pk_size(self.h)
        out = np.zeros(n, dtype=np.uint8)
        lib().orc_pk_serialize(self.h, _ptr(out))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pk_free(self.h)
            self.h = None


def prove(circuit, pk, z, r, s, threads=0):
    """Returns (proof_raw bytes, timings dict)."""
    nb = FQ_BYTES[circuit.curve_id]
    out = np.zeros(8 * nb + 3, dtype=np.uint8)
    tm = np.zeros(8, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.uint8)
    rc = lib().orc_prove(circuit.h, pk.h, _ptr(z), int(r).to_bytes(32, "little"), int(s).to_bytes(32, "little"), _ptr(out),
                         threads or hw_threads(), _ptr(tm))
    if rc != 0:
        raise RuntimeError("oracle prove failed (shape mismatch)")
    names = ("matvec", "fft", "msm_h", "msm_l", "msm_a", "msm_b1", "msm_b2", "total")
    return out.tobytes(), dict(zip(names, tm.tolist()))


def trapdoor(circuit, toxic, z, r, s, threads=0):
    nb = FQ_BYTES[circuit.curve_id]
    out = np.zeros(8 * nb + 3, dtype=np.uint8)
    z = np.ascontiguousarray(z, dtype=np.uint8)
    lib().orc_trapdoor(circuit.h, toxic, _ptr(z), int(r).to_bytes(32, "little"), int(s).to_bytes(32, "little"), _ptr(out),
                       threads or hw_threads())
    return out.tobytes()


def witness_map(circuit, z, threads=0):
    out = np.zeros(circuit.N * 32, dtype=np.uint8)
    z = np.ascontiguousarray(z, dtype=np.uint8)
    lib().orc_witness_map(circuit.h, _ptr(z), _ptr(out), threads or hw_threads())
    return out


def ntt(curve_id, data, direction, threads=0):
    """direction: 'fft' | 'ifft' | 'coset_fft' | 'coset_ifft'; data uint8[N*32] canonical LE; returns new array."""
    d = np.array(data, dtype=np.uint8, copy=True)
    n = d.size // 32
    logn = n.bit_length() - 1
    assert 1 << logn == n
    code = {"fft": 0, "ifft": 1, "coset_fft": 2, "coset_ifft": 3}[direction]
    lib().orc_ntt(curve_id, logn, code, _ptr(d), threads or hw_threads())
    return d


def msm(curve_id, group, bases, scalars, threads=0):
    nb = FQ_BYTES[curve_id]
    pt = 2 * nb * group
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
    n = scalars.size // 32
    assert bases.size == n * pt
    out = np.zeros(pt + 1, dtype=np.uint8)
    lib().orc_msm(curve_id, group, n, _ptr(bases), _ptr(scalars), _ptr(out), threads or hw_threads())
    return out.tobytes()


def field_op(curve_id, field, op, a, b=0):
    nb = 32 if field == 0 else FQ_BYTES[curve_id]
    out = np.zeros(nb, dtype=np.uint8)
    code = {"add": 0, "sub": 1, "mul": 2, "inv": 3}[op]
    lib().orc_field_op(curve_id, field, code, int(a).to_bytes(nb, "little"), int(b).to_bytes(nb, "little"), _ptr(out))
    return int.from_bytes(out.tobytes(), "little")


def generators(curve_id):
    nb = FQ_BYTES[curve_id]
    g1 = np.zeros(2 * nb, dtype=np.uint8)
    g2 = np.zeros(4 * nb, dtype=np.uint8)
    lib().orc_generators(curve_id, _ptr(g1), _ptr(g2))
    return g1.tobytes(), g2.tobytes()


# ---------------- GM17 (oracle/c/gm17.hpp) ----------------
def gm17_toxic_bytes(tox):
    return b"".join(int(v).to_bytes(32, "little") for v in (tox.alpha, tox.beta, tox.gamma, tox.t))


class Gm17ProvingKey:
    def __init__(self, handle, curve_id):
        self.h = handle
        self.curve_id = curve_id

    @staticmethod
    def setup(circuit, toxic, threads=0):
        return Gm17ProvingKey(lib().orc_gm17_setup(circuit.h, toxic, threads or hw_threads()), circuit.curve_id)

    @staticmethod
    def parse(curve_id, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        h = lib().orc_gm17_pk_parse(curve_id, _ptr(data), data.size)
        if not h:
            raise ValueError("malformed GM17 proving key")
        return Gm17ProvingKey(h, curve_id)

    def serialize(self):
        n = lib().orc_gm17_pk_size(self.h)
        out = np.zeros(n, dtype=np.uint8)
        lib().orc_gm17_pk_serialize(self.h, _ptr(out))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pk_free(self.h)
            self.h = None


def gm17_prove(circuit, pk, z, d1, d2, r, threads=0):
    """ark-gm17's create_proof, term by term.  Returns (proof_raw bytes, timings dict)."""
    nb = FQ_BYTES[circuit.curve_id]
    out = np.zeros(8 * nb + 3, dtype=np.uint8)
    tm = np.zeros(8, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.uint8)
    b = lambda v: int(v).to_bytes(32, "little")
    rc = lib().orc_gm17_prove(circuit.h, pk.h, _ptr(z), b(d1), b(d2), b(r), _ptr(out), threads or hw_threads(), _ptr(tm))
    if rc != 0:
        raise RuntimeError("oracle GM17 prove failed (shape mismatch)")
    names = ("matvec", "witness_map", "msm_g", "msm_c", "msm_a", "msm_b1", "msm_b", "total")
    return out.tobytes(), dict(zip(names, tm.tolist()))


def gm17_trapdoor(circuit, toxic, z, d1, r, threads=0):
    nb = FQ_BYTES[circuit.curve_id]
    out = np.zeros(8 * nb + 3, dtype=np.uint8)
    z = np.ascontiguousarray(z, dtype=np.uint8)
    b = lambda v: int(v).to_bytes(32, "little")
    lib().orc_gm17_trapdoor(circuit.h, toxic, _ptr(z), b(d1), b(r), _ptr(out), threads or hw_threads())
    return out.tobytes()
