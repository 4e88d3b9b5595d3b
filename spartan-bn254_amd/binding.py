"""ctypes binding of include/sbn254.h.  Device pointers are plain ints (torch `tensor.data_ptr()`)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SBN_SCALARS_MONT = 1
SBN_POINTS_MONT = 2

EXPORTED_SYMBOLS = [
    "sbn_ctx_create", "sbn_ctx_destroy", "sbn_last_error", "sbn_ctx_set_stream", "sbn_ctx_sync", "sbn_version",
    "sbn_dev_alloc", "sbn_dev_free", "sbn_dev_upload", "sbn_dev_download",
    "sbn_msm", "sbn_msm_jacobian", "sbn_bases_split_at", "sbn_bases_scale", "sbn_bases_upload", "sbn_bases_precompute", "sbn_bases_free", "sbn_bases_len", "sbn_gens_new", "sbn_bases_synthetic", "sbn_scalars_synthetic", "sbn_bases_download",
    "sbn_msm_bases", "sbn_msm_bases_dev", "sbn_commit_rows", "sbn_commit_rows_dev", "sbn_g1_compress", "sbn_g1_sum", "sbn_unipoly_from_evals", "sbn_unipoly_eval", "sbn_factored_lens",
    "sbn_table_upload", "sbn_table_from_dev", "sbn_table_free", "sbn_table_len", "sbn_table_download", "sbn_table_read0", "sbn_table_read0_many",
    "sbn_bind_top", "sbn_bind_top_many", "sbn_sc_eval_cubic", "sbn_sc_eval_cubic_batched", "sbn_sc_eval_r1cs", "sbn_sc_eval_quad",
    "sbn_sc_bind_eval_cubic_batched", "sbn_sc_bind_eval_r1cs", "sbn_sc_bind_eval_quad",
    "sbn_sumcheck_begin", "sbn_sumcheck_begin_eq", "sbn_sumcheck_round", "sbn_sumcheck_len", "sbn_sumcheck_finish", "sbn_sumcheck_free",
    "sbn_group_create", "sbn_group_destroy", "sbn_group_size", "sbn_group_ctx", "sbn_group_last_error", "sbn_group_bases_upload", "sbn_group_gens_new", "sbn_group_bases_precompute",
    "sbn_group_bases_free", "sbn_group_commit_rows", "sbn_group_commit_rows_dev", "sbn_group_gather_commit", "sbn_group_msm", "sbn_group_bases_upload_ranges", "sbn_group_bases_synthetic_ranges", "sbn_group_range", "sbn_group_msm_bases", "sbn_group_msm_bases_dev",
    "sbn_eq_evals", "sbn_hash_layer", "sbn_hash_layer_pair", "sbn_product_layer", "sbn_product_circuit", "sbn_product_circuit_many", "sbn_table_halves", "sbn_table_slice", "sbn_table_dot", "sbn_table_evaluate", "sbn_table_evaluate_many", "sbn_table_bound", "sbn_gather_merge", "sbn_gather_merge_rows", "sbn_commit_table", "sbn_bullet_begin", "sbn_bullet_begin_scaled", "sbn_bullet_free", "sbn_bullet_len", "sbn_bullet_cross", "sbn_bullet_fold_cross", "sbn_bullet_fold", "sbn_bullet_finish", "sbn_prof_enable", "sbn_prof_reset", "sbn_prof_count", "sbn_prof_get", "sbn_prof_last_job",
]


class SbnError(RuntimeError):
    pass


def lib_path():
    return os.path.join(_HERE, "libsbn254_hip.so")


def build_library():
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return lib_path()


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7, like /opt/rocm's).  Import it FIRST so that this
        # library binds to the HIP runtime torch already loaded; the other order puts two runtimes in one process and
        # the second one finds no GPU.  Without torch (the Rust shim) /opt/rocm's runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        p = lib_path()
        if not os.path.exists(p):
            raise SbnError(f"{p} is missing: run `make -C {_HERE}` (or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(p)
        L.sbn_last_error.restype = C.c_char_p
        L.sbn_version.restype = C.c_char_p
        missing = [s for s in EXPORTED_SYMBOLS if not hasattr(L, s)]
        if missing:
            raise SbnError(f"{p} does not export {missing}; rebuild it")
        L.sbn_bases_len.restype = C.c_size_t
        L.sbn_table_len.restype = C.c_size_t
        L.sbn_bullet_len.restype = C.c_size_t
        L.sbn_sumcheck_len.restype = C.c_size_t
        L.sbn_group_size.restype = C.c_size_t
        L.sbn_group_ctx.restype = C.c_void_p
        L.sbn_group_last_error.restype = C.c_char_p
        L.sbn_factored_lens.restype = None
        for name in ("sbn_ctx_destroy", "sbn_bases_free", "sbn_table_free", "sbn_bullet_free", "sbn_sumcheck_free", "sbn_group_destroy", "sbn_group_bases_free", "sbn_group_range"):
            getattr(L, name).restype = None
        _LIB = L
    return _LIB


HARNESS_STAGES = ["r1cs_sat_proof", "eq_evals", "derefs_computation", "derefs_commitment", "network_construction", "network_proof",
                  "sub_witness_commit", "sub_phase1_sumcheck", "sub_phase2_sumcheck", "sub_witness_opening", "sub_ops_sumchecks", "sub_mem_sumchecks",
                  "sub_evaluations", "sub_openings"]


class HarnessParams(C.Structure):
    _fields_ = [("log_ops", C.c_int32), ("log_mem", C.c_int32), ("log_cons", C.c_int32), ("stateful_sumcheck", C.c_int32),
                ("lookup_bytes_sat", C.c_uint64), ("lookup_bytes_eval", C.c_uint64), ("seed", C.c_uint64), ("rounds_out", C.c_uint32 * 4),
                ("passes", C.c_uint32), ("trace_markers", C.c_uint32)]


_HARNESS = None


def harness_lib():
    """libsbn_prove_harness.so: the compiled caller of the C ABI (harness/prove_stages.cpp)"""
    global _HARNESS
    if _HARNESS is None:
        lib()                                                  # the product library first (the harness links against it)
        p = os.path.join(_HERE, "libsbn_prove_harness.so")
        if not os.path.exists(p):
            raise SbnError(f"{p} is missing: run `make -C {_HERE}`")
        _HARNESS = C.CDLL(p)
    return _HARNESS


def harness_prove(ctx, log_ops, log_mem, log_cons, stateful=True, lookup_bytes_sat=0, lookup_bytes_eval=0, seed=1, trace_cap=0, passes=1, trace_markers=False):
    """a keyless-shaped prove's device-side stages from compiled code -> (stage_ms dict, digest, trace bytes, rounds dict); with passes > 1
    the proves run back to back on one setup and the fastest pass's times are returned (the trace is the last pass's)"""
    prm = HarnessParams(log_ops, log_mem, log_cons, 1 if stateful else 0, lookup_bytes_sat, lookup_bytes_eval, seed)
    prm.passes = passes
    prm.trace_markers = 1 if trace_markers else 0
    ms = (C.c_double * 16)(); dig = (C.c_uint8 * 32)(); err = C.create_string_buffer(512)
    tr = (C.c_uint8 * trace_cap)() if trace_cap else None; tl = C.c_size_t(0)
    rc = harness_lib().sbn_harness_prove(ctx.h, C.byref(prm), ms, dig, tr, C.c_size_t(trace_cap), C.byref(tl), err, C.c_size_t(512))
    if rc:
        raise SbnError("sbn_harness_prove: " + err.value.decode())
    stages = {n: ms[i] for i, n in enumerate(HARNESS_STAGES)}
    rounds = {"sumcheck_rounds_ops": prm.rounds_out[0], "sumcheck_rounds_mem": prm.rounds_out[1], "bullet_rounds": prm.rounds_out[2], "layers": prm.rounds_out[3]}
    return stages, bytes(dig), (bytes(tr[:tl.value]) if trace_cap else b""), rounds


def _ptr(x):
    """host bytes-like -> char pointer; None -> NULL"""
    if x is None:
        return None
    if isinstance(x, (bytes, bytearray)):
        return (C.c_uint8 * len(x)).from_buffer_copy(x) if isinstance(x, bytes) else (C.c_uint8 * len(x)).from_buffer(x)
    if hasattr(x, "ctypes"):  # numpy array
        return x.ctypes.data_as(C.POINTER(C.c_uint8))
    raise TypeError(type(x))


def g1_compress(xy):
    n = len(xy) // 64
    out = (C.c_uint8 * (32 * n))()
    rc = lib().sbn_g1_compress(_ptr(xy), C.c_size_t(n), out)
    if rc:
        raise SbnError(f"sbn_g1_compress rc={rc}")
    return bytes(out)


def g1_sum(xy):
    """host-side sum of canonical affine points (the fold after the all-gather of per-GPU partials)"""
    n = len(xy) // 64
    out = (C.c_uint8 * 64)(); inf = C.c_int()
    rc = lib().sbn_g1_sum(_ptr(xy), C.c_size_t(n), out, C.byref(inf))
    if rc:
        raise SbnError(f"sbn_g1_sum rc={rc}")
    return bytes(out), bool(inf.value)


def unipoly_from_evals(evals):
    """UniPoly::from_evals (unipoly.rs:28-59), host side"""
    n = len(evals) // 32; out = (C.c_uint8 * (32 * n))()
    rc = lib().sbn_unipoly_from_evals(_ptr(evals), C.c_size_t(n), out)
    if rc:
        raise SbnError(f"sbn_unipoly_from_evals rc={rc}")
    return bytes(out)


def unipoly_eval(coeffs, r):
    out = (C.c_uint8 * 32)()
    rc = lib().sbn_unipoly_eval(_ptr(coeffs), C.c_size_t(len(coeffs) // 32), _ptr(r), out)
    if rc:
        raise SbnError(f"sbn_unipoly_eval rc={rc}")
    return bytes(out)


def factored_lens(ell):
    a, b = C.c_size_t(), C.c_size_t()
    lib().sbn_factored_lens(C.c_size_t(ell), C.byref(a), C.byref(b))
    return a.value, b.value


class Bases:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def __len__(self):
        return lib().sbn_bases_len(self.h)

    def free(self):
        if self.h:
            lib().sbn_bases_free(self.ctx.h, self.h)
            self.h = None


class Bullet:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def __len__(self):
        return lib().sbn_bullet_len(self.h)

    def free(self):
        if self.h:
            lib().sbn_bullet_free(self.ctx.h, self.h)
            self.h = None


class Sumcheck:
    """prove_cubic_batched (sumcheck.rs:165-330) as a device-resident state (sbn_sumcheck_*)"""

    def __init__(self, ctx, handle, n_par, n_seq):
        self.ctx, self.h, self.ntab = ctx, handle, 2 * n_par + (1 if n_par else 0) + 3 * n_seq

    def __len__(self):
        return lib().sbn_sumcheck_len(self.h)

    def round(self, r):
        out = (C.c_uint8 * 96)()
        self.ctx._chk(lib().sbn_sumcheck_round(self.ctx.h, self.h, _ptr(r), out), "sbn_sumcheck_round")
        return bytes(out)

    def finish(self):
        out = (C.c_uint8 * (32 * self.ntab))()
        self.ctx._chk(lib().sbn_sumcheck_finish(self.ctx.h, self.h, out), "sbn_sumcheck_finish")
        return [bytes(out[32 * t:32 * t + 32]) for t in range(self.ntab)]

    def free(self):
        if self.h:
            lib().sbn_sumcheck_free(self.ctx.h, self.h)
            self.h = None


class Table:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def __len__(self):
        return lib().sbn_table_len(self.h)

    def free(self):
        if self.h:
            lib().sbn_table_free(self.ctx.h, self.h)
            self.h = None


class GroupBases:
    def __init__(self, group, handle):
        self.group, self.h = group, handle

    def range(self, device):
        lo, hi = C.c_size_t(), C.c_size_t()
        lib().sbn_group_range(self.h, C.c_size_t(device), C.byref(lo), C.byref(hi)); return lo.value, hi.value

    def free(self):
        if self.h:
            lib().sbn_group_bases_free(self.group.h, self.h); self.h = None


class Group:
    """sbn_group_*: one process, several devices behind one call (a device may be listed more than once)"""

    def __init__(self, devices):
        self.h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        rc = lib().sbn_group_create(arr, C.c_size_t(len(devices)), C.byref(self.h))
        if rc:
            raise SbnError(f"sbn_group_create failed rc={rc}")

    def close(self):
        if self.h:
            lib().sbn_group_destroy(self.h); self.h = None

    def __len__(self):
        return lib().sbn_group_size(self.h)

    def _chk(self, rc, what):
        if rc:
            raise SbnError(f"{what}: rc={rc}: {lib().sbn_group_last_error(self.h).decode()}")

    def ctx(self, i):
        """the i-th device's context as a (non-owning) Context"""
        c = Context.__new__(Context); c.h = C.c_void_p(lib().sbn_group_ctx(self.h, C.c_size_t(i))); c.owned = False
        return c

    def bases_upload(self, G_xy, h_xy=None, flags=0):
        o = C.c_void_p()
        self._chk(lib().sbn_group_bases_upload(self.h, _ptr(G_xy), C.c_size_t(len(G_xy) // 64), _ptr(h_xy), C.c_uint32(flags), C.byref(o)), "sbn_group_bases_upload")
        return GroupBases(self, o)

    def gens_new(self, n, label, want_points=True):
        o = C.c_void_p(); out = (C.c_uint8 * (64 * (n + 1)))() if want_points else None
        self._chk(lib().sbn_group_gens_new(self.h, C.c_size_t(n), _ptr(label), C.c_size_t(len(label)), out, C.byref(o)), "sbn_group_gens_new")
        return GroupBases(self, o), (bytes(out) if want_points else None)

    def bases_precompute(self, gb, max_bytes_per_device):
        cw = C.c_int(0)
        self._chk(lib().sbn_group_bases_precompute(self.h, gb.h, C.c_size_t(max_bytes_per_device), C.byref(cw)), "sbn_group_bases_precompute"); return cw.value

    def commit_rows(self, gb, Z, blinds, L, R, flags=0):
        out = (C.c_uint8 * (64 * L))(); inf = (C.c_uint8 * L)()
        self._chk(lib().sbn_group_commit_rows(self.h, gb.h, _ptr(Z), _ptr(blinds), C.c_size_t(L), C.c_size_t(R), C.c_uint32(flags), out, inf), "sbn_group_commit_rows")
        return bytes(out), bytes(inf)

    def commit_rows_dev(self, gb, z_ptrs, blind_ptrs, L, R, flags=0):
        """z_ptrs[d]: device pointer on device d to its interleaved rows (d, d + N, ...), blind_ptrs likewise or None"""
        N = len(self)
        za = (C.c_void_p * N)(*z_ptrs); ba = (C.c_void_p * N)(*blind_ptrs) if blind_ptrs is not None else None
        out = (C.c_uint8 * (64 * L))(); inf = (C.c_uint8 * L)()
        self._chk(lib().sbn_group_commit_rows_dev(self.h, gb.h, za, ba, C.c_size_t(L), C.c_size_t(R), C.c_uint32(flags), out, inf), "sbn_group_commit_rows_dev")
        return bytes(out), bytes(inf)

    def gather_commit(self, gb, mem, addr_ptrs, n, L, R):
        """mem[d][k]: Table on device d, addr_ptrs[d][k]: device pointer to its n uint32 addresses -> (L x 64 B, L flags)"""
        N = len(self); count = len(mem[0])
        ma = (C.c_void_p * (N * count))(*[t.h for row in mem for t in row])
        aa = (C.c_void_p * (N * count))(*[a for row in addr_ptrs for a in row])
        out = (C.c_uint8 * (64 * L))(); inf = (C.c_uint8 * L)()
        self._chk(lib().sbn_group_gather_commit(self.h, gb.h, ma, aa, C.c_size_t(count), C.c_size_t(n), C.c_size_t(L), C.c_size_t(R), out, inf), "sbn_group_gather_commit")
        return bytes(out), bytes(inf)

    def msm(self, scalars, points, flags=0):
        out = (C.c_uint8 * 64)(); inf = C.c_int()
        self._chk(lib().sbn_group_msm(self.h, _ptr(scalars), _ptr(points), C.c_size_t(len(scalars) // 32), C.c_uint32(flags), out, C.byref(inf)), "sbn_group_msm")
        return bytes(out), bool(inf.value)

    def bases_upload_ranges(self, G_xy, flags=0):
        o = C.c_void_p()
        self._chk(lib().sbn_group_bases_upload_ranges(self.h, _ptr(G_xy), C.c_size_t(len(G_xy) // 64), C.c_uint32(flags), C.byref(o)), "sbn_group_bases_upload_ranges")
        return GroupBases(self, o)

    def bases_synthetic_ranges(self, n, s0, d):
        o = C.c_void_p()
        self._chk(lib().sbn_group_bases_synthetic_ranges(self.h, C.c_size_t(n), _ptr(s0), _ptr(d), C.byref(o)), "sbn_group_bases_synthetic_ranges")
        return GroupBases(self, o)

    def msm_bases(self, gb, scalars, flags=0):
        out = (C.c_uint8 * 64)(); inf = C.c_int()
        self._chk(lib().sbn_group_msm_bases(self.h, gb.h, _ptr(scalars), C.c_size_t(len(scalars) // 32), C.c_uint32(flags), out, C.byref(inf)), "sbn_group_msm_bases")
        return bytes(out), bool(inf.value)

    def msm_bases_dev(self, gb, dev_ptrs, flags=0):
        arr = (C.c_void_p * len(dev_ptrs))(*dev_ptrs); out = (C.c_uint8 * 64)(); inf = C.c_int()
        self._chk(lib().sbn_group_msm_bases_dev(self.h, gb.h, arr, C.c_uint32(flags), out, C.byref(inf)), "sbn_group_msm_bases_dev")
        return bytes(out), bool(inf.value)


class Context:
    owned = True

    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = lib().sbn_ctx_create(device, C.byref(self.h))
        if rc:
            raise SbnError(f"sbn_ctx_create failed rc={rc} (no gfx950 device? there is no CPU fallback)")

    def close(self):
        if self.h and self.owned:
            lib().sbn_ctx_destroy(self.h)
        self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc:
            raise SbnError(f"{what}: rc={rc}: {lib().sbn_last_error(self.h).decode()}")

    def set_stream(self, stream_ptr):
        self._chk(lib().sbn_ctx_set_stream(self.h, C.c_void_p(stream_ptr)), "set_stream")

    def sync(self):
        self._chk(lib().sbn_ctx_sync(self.h), "sync")

    # ---- B1
    def msm(self, scalars, points, flags=0):
        n = len(scalars) // 32
        assert len(points) == 64 * n
        out = (C.c_uint8 * 64)(); inf = C.c_int()
        self._chk(lib().sbn_msm(self.h, _ptr(scalars), _ptr(points), C.c_size_t(n), C.c_uint32(flags), out, C.byref(inf)), "sbn_msm")
        return bytes(out), bool(inf.value)

    def msm_jacobian(self, scalars, points_xyz, flags=0):
        n = len(scalars) // 32
        assert len(points_xyz) == 96 * n
        out = (C.c_uint8 * 64)(); inf = C.c_int()
        self._chk(lib().sbn_msm_jacobian(self.h, _ptr(scalars), _ptr(points_xyz), C.c_size_t(n), C.c_uint32(flags), out, C.byref(inf)), "sbn_msm_jacobian")
        return bytes(out), bool(inf.value)

    def bases_precompute(self, bases, max_bytes):
        """build the fixed-base lookup table within max_bytes of HBM; returns the window bits chosen"""
        cw = C.c_int(0)
        self._chk(lib().sbn_bases_precompute(self.h, bases.h, C.c_size_t(max_bytes), C.byref(cw)), "sbn_bases_precompute")
        return cw.value

    def bases_split_at(self, bases, mid):
        l, r = C.c_void_p(), C.c_void_p()
        self._chk(lib().sbn_bases_split_at(self.h, bases.h, C.c_size_t(mid), C.byref(l), C.byref(r)), "sbn_bases_split_at")
        return Bases(self, l), Bases(self, r)

    def bases_scale(self, bases, s):
        o = C.c_void_p()
        self._chk(lib().sbn_bases_scale(self.h, bases.h, _ptr(s), C.byref(o)), "sbn_bases_scale")
        return Bases(self, o)

    def bases_upload(self, G_xy, h_xy=None, flags=0):
        n = len(G_xy) // 64
        hb = C.c_void_p()
        self._chk(lib().sbn_bases_upload(self.h, _ptr(G_xy), C.c_size_t(n), _ptr(h_xy), C.c_uint32(flags), C.byref(hb)), "sbn_bases_upload")
        return Bases(self, hb)

    def gens_new(self, n, label, want_points=True):
        hb = C.c_void_p()
        out = (C.c_uint8 * (64 * (n + 1)))() if want_points else None
        self._chk(lib().sbn_gens_new(self.h, C.c_size_t(n), _ptr(label), C.c_size_t(len(label)), out, C.byref(hb)), "sbn_gens_new")
        return Bases(self, hb), (bytes(out) if want_points else None)

    def bases_synthetic(self, n, first, s0, d):
        hb = C.c_void_p()
        self._chk(lib().sbn_bases_synthetic(self.h, C.c_size_t(n), C.c_uint64(first), _ptr(s0), _ptr(d), C.byref(hb)), "sbn_bases_synthetic")
        return Bases(self, hb)

    def scalars_synthetic(self, seed, first, n, out_dev_ptr):
        self._chk(lib().sbn_scalars_synthetic(self.h, C.c_uint64(seed), C.c_uint64(first), C.c_size_t(n), C.c_void_p(out_dev_ptr)), "sbn_scalars_synthetic")

    def bases_download(self, bases, first, count):
        out = (C.c_uint8 * (64 * count))()
        self._chk(lib().sbn_bases_download(self.h, bases.h, C.c_size_t(first), C.c_size_t(count), out), "sbn_bases_download")
        return bytes(out)

    def msm_bases(self, bases, scalars, flags=0):
        n = len(scalars) // 32
        out = (C.c_uint8 * 64)(); inf = C.c_int()
        self._chk(lib().sbn_msm_bases(self.h, bases.h, _ptr(scalars), C.c_size_t(n), C.c_uint32(flags), out, C.byref(inf)), "sbn_msm_bases")
        return bytes(out), bool(inf.value)

    def msm_bases_dev(self, bases, scalars_dev_ptr, n, flags=0):
        out = (C.c_uint8 * 64)(); inf = C.c_int()
        self._chk(lib().sbn_msm_bases_dev(self.h, bases.h, C.c_void_p(scalars_dev_ptr), C.c_size_t(n), C.c_uint32(flags), out, C.byref(inf)), "sbn_msm_bases_dev")
        return bytes(out), bool(inf.value)

    # ---- B2
    def commit_rows(self, bases, Z, blinds, L, R, flags=0):
        out = (C.c_uint8 * (64 * L))(); inf = (C.c_uint8 * L)()
        self._chk(lib().sbn_commit_rows(self.h, bases.h, _ptr(Z), _ptr(blinds), C.c_size_t(L), C.c_size_t(R), C.c_uint32(flags), out, inf), "sbn_commit_rows")
        return bytes(out), bytes(inf)

    def commit_rows_dev(self, bases, Z_dev_ptr, blinds_dev_ptr, L, R, flags=0):
        out = (C.c_uint8 * (64 * L))(); inf = (C.c_uint8 * L)()
        self._chk(lib().sbn_commit_rows_dev(self.h, bases.h, C.c_void_p(Z_dev_ptr), C.c_void_p(blinds_dev_ptr or 0), C.c_size_t(L), C.c_size_t(R), C.c_uint32(flags), out, inf), "sbn_commit_rows_dev")
        return bytes(out), bytes(inf)

    # ---- raw device memory
    def dev_alloc(self, nbytes):
        p = C.c_void_p(); self._chk(lib().sbn_dev_alloc(self.h, C.c_size_t(nbytes), C.byref(p)), "sbn_dev_alloc"); return p.value

    def dev_free(self, p):
        self._chk(lib().sbn_dev_free(self.h, C.c_void_p(p)), "sbn_dev_free")

    def dev_upload(self, dst, data):
        self._chk(lib().sbn_dev_upload(self.h, C.c_void_p(dst), _ptr(data), C.c_size_t(len(data) if not hasattr(data, "nbytes") else data.nbytes)), "sbn_dev_upload")

    def dev_download(self, src, nbytes):
        out = (C.c_uint8 * nbytes)(); self._chk(lib().sbn_dev_download(self.h, out, C.c_void_p(src), C.c_size_t(nbytes)), "sbn_dev_download"); return bytes(out)

    # ---- B3
    def table_upload(self, Z, flags=0):
        n = len(Z) // 32; ht = C.c_void_p()
        self._chk(lib().sbn_table_upload(self.h, _ptr(Z), C.c_size_t(n), C.c_uint32(flags), C.byref(ht)), "sbn_table_upload")
        return Table(self, ht)

    def table_from_dev(self, dev_ptr, n, flags=0):
        ht = C.c_void_p()
        self._chk(lib().sbn_table_from_dev(self.h, C.c_void_p(dev_ptr), C.c_size_t(n), C.c_uint32(flags), C.byref(ht)), "sbn_table_from_dev")
        return Table(self, ht)

    def table_download(self, t):
        n = len(t); out = (C.c_uint8 * (32 * n))()
        self._chk(lib().sbn_table_download(self.h, t.h, out), "sbn_table_download"); return bytes(out)

    def table_read0(self, t):
        out = (C.c_uint8 * 32)(); self._chk(lib().sbn_table_read0(self.h, t.h, out), "sbn_table_read0"); return bytes(out)

    def table_read0_many(self, ts):
        """entry 0 of every table in `ts`: one launch, one wait (sbn_table_read0_many)"""
        n = len(ts)
        arr = (C.c_void_p * max(n, 1))(*[t.h for t in ts]); out = (C.c_uint8 * (32 * max(n, 1)))()
        self._chk(lib().sbn_table_read0_many(self.h, arr, C.c_size_t(n), out), "sbn_table_read0_many")
        return [bytes(out[32 * i:32 * i + 32]) for i in range(n)]

    def bind_top(self, t, r):
        self._chk(lib().sbn_bind_top(self.h, t.h, _ptr(r)), "sbn_bind_top")

    def bind_top_many(self, ts, r):
        arr = (C.c_void_p * len(ts))(*[t.h for t in ts])
        self._chk(lib().sbn_bind_top_many(self.h, arr, C.c_size_t(len(ts)), _ptr(r)), "sbn_bind_top_many")

    def sc_eval_cubic(self, A, B, Cc):
        out = (C.c_uint8 * 96)(); self._chk(lib().sbn_sc_eval_cubic(self.h, A.h, B.h, Cc.h, out), "sbn_sc_eval_cubic"); return bytes(out)

    def sc_eval_cubic_batched(self, As, Bs, Cs):
        k = len(As); mk = lambda ts: (C.c_void_p * k)(*[t.h for t in ts])
        out = (C.c_uint8 * (96 * k))()
        self._chk(lib().sbn_sc_eval_cubic_batched(self.h, mk(As), mk(Bs), mk(Cs), C.c_size_t(k), out), "sbn_sc_eval_cubic_batched"); return bytes(out)

    def sc_eval_r1cs(self, T, A, B, Cc):
        out = (C.c_uint8 * 96)(); self._chk(lib().sbn_sc_eval_r1cs(self.h, T.h, A.h, B.h, Cc.h, out), "sbn_sc_eval_r1cs"); return bytes(out)

    def sc_eval_quad(self, Z, ABC):
        out = (C.c_uint8 * 64)(); self._chk(lib().sbn_sc_eval_quad(self.h, Z.h, ABC.h, out), "sbn_sc_eval_quad"); return bytes(out)

    def sc_bind_eval_cubic_batched(self, As, Bs, Cs, r):
        k = len(As); mk = lambda ts: (C.c_void_p * k)(*[t.h for t in ts])
        out = (C.c_uint8 * (96 * k))()
        self._chk(lib().sbn_sc_bind_eval_cubic_batched(self.h, mk(As), mk(Bs), mk(Cs), C.c_size_t(k), _ptr(r), out), "sbn_sc_bind_eval_cubic_batched"); return bytes(out)

    def sc_bind_eval_r1cs(self, T, A, B, Cc, r):
        out = (C.c_uint8 * 96)(); self._chk(lib().sbn_sc_bind_eval_r1cs(self.h, T.h, A.h, B.h, Cc.h, _ptr(r), out), "sbn_sc_bind_eval_r1cs"); return bytes(out)

    def sc_bind_eval_quad(self, Z, ABC, r):
        out = (C.c_uint8 * 64)(); self._chk(lib().sbn_sc_bind_eval_quad(self.h, Z.h, ABC.h, _ptr(r), out), "sbn_sc_bind_eval_quad"); return bytes(out)

    def sumcheck_begin(self, A_par, B_par, C_par, A_seq, B_seq, C_seq, coeffs):
        """-> (Sumcheck state, the combined (e0, e2, e3) of round 0)"""
        mk = lambda ts: (C.c_void_p * max(1, len(ts)))(*[t.h for t in ts])
        st = C.c_void_p(); out = (C.c_uint8 * 96)()
        self._chk(lib().sbn_sumcheck_begin(self.h, mk(A_par), mk(B_par), C_par.h if C_par is not None else None, C.c_size_t(len(A_par)),
                                           mk(A_seq), mk(B_seq), mk(C_seq), C.c_size_t(len(A_seq)), _ptr(coeffs), out, C.byref(st)), "sbn_sumcheck_begin")
        return Sumcheck(self, st, len(A_par), len(A_seq)), bytes(out)

    def sumcheck_begin_eq(self, A_par, B_par, rand, A_seq, B_seq, C_seq, coeffs):
        """poly_C_par = eq(rand) built inside the call -> (Sumcheck state, the combined (e0, e2, e3) of round 0)"""
        mk = lambda ts: (C.c_void_p * max(1, len(ts)))(*[t.h for t in ts])
        st = C.c_void_p(); out = (C.c_uint8 * 96)()
        self._chk(lib().sbn_sumcheck_begin_eq(self.h, mk(A_par), mk(B_par), C.c_size_t(len(A_par)), _ptr(rand), C.c_size_t(len(rand) // 32),
                                              mk(A_seq), mk(B_seq), mk(C_seq), C.c_size_t(len(A_seq)), _ptr(coeffs), out, C.byref(st)), "sbn_sumcheck_begin_eq")
        return Sumcheck(self, st, len(A_par), len(A_seq)), bytes(out)

    def eq_evals(self, r):
        ell = len(r) // 32; ht = C.c_void_p()
        self._chk(lib().sbn_eq_evals(self.h, _ptr(r), C.c_size_t(ell), C.byref(ht)), "sbn_eq_evals"); return Table(self, ht)

    def hash_layer(self, addr_dev_ptr, val, ts_dev_ptr, ts_add, r_hash, r_multiset):
        ht = C.c_void_p()
        self._chk(lib().sbn_hash_layer(self.h, C.c_void_p(addr_dev_ptr or 0), val.h, C.c_void_p(ts_dev_ptr or 0), C.c_uint32(ts_add), _ptr(r_hash), _ptr(r_multiset), C.byref(ht)), "sbn_hash_layer")
        return Table(self, ht)

    def hash_layer_pair(self, addr_dev_ptr, val, ts_a_ptr, ts_a_add, ts_b_ptr, ts_b_add, r_hash, r_multiset):
        ha, hb = C.c_void_p(), C.c_void_p()
        self._chk(lib().sbn_hash_layer_pair(self.h, C.c_void_p(addr_dev_ptr or 0), val.h, C.c_void_p(ts_a_ptr or 0), C.c_uint32(ts_a_add), C.c_void_p(ts_b_ptr or 0), C.c_uint32(ts_b_add),
                                            _ptr(r_hash), _ptr(r_multiset), C.byref(ha), C.byref(hb)), "sbn_hash_layer_pair")
        return Table(self, ha), Table(self, hb)

    def product_layer(self, t):
        ht = C.c_void_p(); self._chk(lib().sbn_product_layer(self.h, t.h, C.byref(ht)), "sbn_product_layer"); return Table(self, ht)

    def product_circuit(self, t):
        """all layers above t (ProductCircuit::new): list of Tables of len/2, len/4, ..., 1 entries"""
        cap = max(1, len(t).bit_length())
        arr = (C.c_void_p * cap)(); cnt = C.c_size_t(0)
        self._chk(lib().sbn_product_circuit(self.h, t.h, arr, C.c_size_t(cap), C.byref(cnt)), "sbn_product_circuit")
        return [Table(self, C.c_void_p(arr[i])) for i in range(cnt.value)]

    def product_circuit_many(self, ts):
        """the circuits of several tables of one length, one launch per layer for all -> [[layers of ts[0]], [layers of ts[1]], ...]"""
        n = len(ts); cap = max(1, len(ts[0]).bit_length())
        ins = (C.c_void_p * n)(*[t.h for t in ts]); arr = (C.c_void_p * (n * cap))(); cnt = C.c_size_t(0)
        self._chk(lib().sbn_product_circuit_many(self.h, ins, C.c_size_t(n), arr, C.c_size_t(cap), C.byref(cnt)), "sbn_product_circuit_many")
        return [[Table(self, C.c_void_p(arr[i * cap + k])) for k in range(cnt.value)] for i in range(n)]

    def table_halves(self, t):
        l, r = C.c_void_p(), C.c_void_p()
        self._chk(lib().sbn_table_halves(self.h, t.h, C.byref(l), C.byref(r)), "sbn_table_halves"); return Table(self, l), Table(self, r)

    def table_dot(self, a, b):
        out = (C.c_uint8 * 32)(); self._chk(lib().sbn_table_dot(self.h, a.h, b.h, out), "sbn_table_dot"); return bytes(out)

    def table_evaluate(self, Z, r):
        out = (C.c_uint8 * 32)(); self._chk(lib().sbn_table_evaluate(self.h, Z.h, _ptr(r), C.c_size_t(len(r) // 32), out), "sbn_table_evaluate"); return bytes(out)

    def table_evaluate_many(self, Zs, r):
        k = len(Zs); arr = (C.c_void_p * k)(*[t.h for t in Zs]); out = (C.c_uint8 * (32 * k))()
        self._chk(lib().sbn_table_evaluate_many(self.h, arr, C.c_size_t(k), _ptr(r), C.c_size_t(len(r) // 32), out), "sbn_table_evaluate_many")
        return bytes(out)

    def table_bound(self, Z, Lvec):
        ht = C.c_void_p(); self._chk(lib().sbn_table_bound(self.h, Z.h, Lvec.h, C.byref(ht)), "sbn_table_bound"); return Table(self, ht)

    def table_slice(self, t, first, length):
        ht = C.c_void_p()
        self._chk(lib().sbn_table_slice(self.h, t.h, C.c_size_t(first), C.c_size_t(length), C.byref(ht)), "sbn_table_slice")
        return Table(self, ht)

    def gather_merge(self, mems, addr_dev_ptrs, n):
        k = len(mems)
        ma = (C.c_void_p * k)(*[t.h for t in mems]); aa = (C.c_void_p * k)(*addr_dev_ptrs)
        ht = C.c_void_p()
        self._chk(lib().sbn_gather_merge(self.h, ma, aa, C.c_size_t(k), C.c_size_t(n), C.byref(ht)), "sbn_gather_merge")
        return Table(self, ht)

    def gather_merge_rows(self, mems, addr_dev_ptrs, n, R, row0, rstep, nrows):
        k = len(mems)
        ma = (C.c_void_p * k)(*[t.h for t in mems]); aa = (C.c_void_p * k)(*addr_dev_ptrs)
        ht = C.c_void_p()
        self._chk(lib().sbn_gather_merge_rows(self.h, ma, aa, C.c_size_t(k), C.c_size_t(n), C.c_size_t(R), C.c_size_t(row0), C.c_size_t(rstep), C.c_size_t(nrows), C.byref(ht)), "sbn_gather_merge_rows")
        return Table(self, ht)

    def commit_table(self, bases, t, blinds, L, R):
        out = (C.c_uint8 * (64 * L))(); inf = (C.c_uint8 * L)()
        self._chk(lib().sbn_commit_table(self.h, bases.h, t.h, _ptr(blinds), C.c_size_t(L), C.c_size_t(R), out, inf), "sbn_commit_table")
        return bytes(out), bytes(inf)

    # ---- BulletReductionProof::prove (nizk/bullet.rs:41-126) as a device-resident state
    def bullet_begin(self, G, Q_xy, a, b, blind=None, want_gamma=True):
        """-> (Bullet state, Gamma_xy or None)"""
        st = C.c_void_p(); gm = (C.c_uint8 * 64)() if want_gamma else None; gi = C.c_int(0)
        self._chk(lib().sbn_bullet_begin(self.h, G.h, _ptr(Q_xy), a.h, b.h, _ptr(blind), gm, C.byref(gi), C.byref(st)), "sbn_bullet_begin")
        return Bullet(self, st), (bytes(gm) if want_gamma else None)

    def bullet_begin_scaled(self, G, Q_base_xy, q_scale, a, b, blind=None, want_gamma=True):
        """Q = q_scale * Q_base -> (Bullet state, Gamma_xy or None)"""
        st = C.c_void_p(); gm = (C.c_uint8 * 64)() if want_gamma else None; gi = C.c_int(0)
        self._chk(lib().sbn_bullet_begin_scaled(self.h, G.h, _ptr(Q_base_xy), _ptr(q_scale), a.h, b.h, _ptr(blind), gm, C.byref(gi), C.byref(st)), "sbn_bullet_begin_scaled")
        return Bullet(self, st), (bytes(gm) if want_gamma else None)

    def bullet_cross(self, st, blind_L=None, blind_R=None):
        """-> (L_xy, L_inf, R_xy, R_inf, c_L, c_R)"""
        Lo, Ro = (C.c_uint8 * 64)(), (C.c_uint8 * 64)(); li, ri = C.c_int(0), C.c_int(0)
        cl, cr = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
        self._chk(lib().sbn_bullet_cross(self.h, st.h, _ptr(blind_L), _ptr(blind_R), Lo, C.byref(li), Ro, C.byref(ri), cl, cr), "sbn_bullet_cross")
        return bytes(Lo), bool(li.value), bytes(Ro), bool(ri.value), bytes(cl), bytes(cr)

    def bullet_fold_cross(self, st, u, u_inv, blind_L=None, blind_R=None):
        """fold with u, then the next round's cross terms -> (L_xy, L_inf, R_xy, R_inf, c_L, c_R)"""
        Lo, Ro = (C.c_uint8 * 64)(), (C.c_uint8 * 64)(); li, ri = C.c_int(0), C.c_int(0)
        cl, cr = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
        self._chk(lib().sbn_bullet_fold_cross(self.h, st.h, _ptr(u), _ptr(u_inv), _ptr(blind_L), _ptr(blind_R), Lo, C.byref(li), Ro, C.byref(ri), cl, cr), "sbn_bullet_fold_cross")
        return bytes(Lo), bool(li.value), bytes(Ro), bool(ri.value), bytes(cl), bytes(cr)

    def bullet_fold(self, st, u, u_inv):
        self._chk(lib().sbn_bullet_fold(self.h, st.h, _ptr(u), _ptr(u_inv)), "sbn_bullet_fold")

    def bullet_finish(self, st):
        """-> (a_hat, b_hat, g_hat_xy)"""
        ah, bh, gh = (C.c_uint8 * 32)(), (C.c_uint8 * 32)(), (C.c_uint8 * 64)(); gi = C.c_int(0)
        self._chk(lib().sbn_bullet_finish(self.h, st.h, ah, bh, gh, C.byref(gi)), "sbn_bullet_finish")
        return bytes(ah), bytes(bh), bytes(gh)

    # ---- profiling
    def prof_enable(self, on=True):
        self._chk(lib().sbn_prof_enable(self.h, int(on)), "prof_enable")

    def prof_last_job(self):
        """{c, W, slots, buckets} of the most recent MSM / row commit on this context"""
        o = (C.c_uint64 * 4)(); self._chk(lib().sbn_prof_last_job(self.h, o), "prof_last_job")
        return {"c": int(o[0]), "W": int(o[1]), "slots": int(o[2]), "buckets": int(o[3])}

    def prof_reset(self):
        self._chk(lib().sbn_prof_reset(self.h), "prof_reset")

    def prof_get(self):
        res = {}
        for i in range(lib().sbn_prof_count(self.h)):
            name = C.c_char_p(); ms = C.c_double(); cnt = C.c_uint64()
            lib().sbn_prof_get(self.h, i, C.byref(name), C.byref(ms), C.byref(cnt))
            res[name.value.decode()] = (ms.value, cnt.value)
        return res
