"""ctypes loader of libmvhdp.so (the C ABI of include/mvhdp.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmvhdp.so")

MAX_M = 8

# every symbol include/mvhdp.h declares
ABI_SYMBOLS = [
    "mvhdp_create", "mvhdp_destroy", "mvhdp_last_error", "mvhdp_version",
    "mvhdp_set_corpus", "mvhdp_set_assignments", "mvhdp_set_view_presence", "mvhdp_get_assignments",
    "mvhdp_set_hyper", "mvhdp_get_alpha", "mvhdp_build_counts", "mvhdp_build_trees",
    "mvhdp_build_inference_trees", "mvhdp_init_assignments_from_trees",
    "mvhdp_get_counts", "mvhdp_set_counts", "mvhdp_get_tree", "mvhdp_get_doc_topic_hist",
    "mvhdp_get_count_histogram", "mvhdp_view_overlap_sums", "mvhdp_model_log_likelihood", "mvhdp_doc_topic_proportions",
    "mvhdp_gamma_doc_statistics", "mvhdp_dp_table_statistics", "mvhdp_antoniak_draws",
    "mvhdp_sweep", "mvhdp_sweep_many", "mvhdp_get_tuning", "mvhdp_set_tuning", "mvhdp_plan_probe", "mvhdp_tuner_probe",
    "mvhdp_apply_delta", "mvhdp_apply_delta_begin", "mvhdp_apply_delta_rows", "mvhdp_apply_delta_end",
    "mvhdp_trees_current", "mvhdp_get_view_weights",
    "mvhdp_device_buffer", "mvhdp_counts_written", "mvhdp_set_stream", "mvhdp_synchronize",
    "mvhdp_group_create", "mvhdp_group_unique_id", "mvhdp_group_create_rank", "mvhdp_group_destroy", "mvhdp_group_last_error",
    "mvhdp_group_get_info", "mvhdp_group_set_exchange_chunks", "mvhdp_group_build_counts", "mvhdp_group_sweep", "mvhdp_group_abort", "mvhdp_group_drain",
    "mvhdp_group_set_hyper", "mvhdp_group_log_likelihood", "mvhdp_group_doc_topic_hist", "mvhdp_group_count_histogram",
    "mvhdp_group_view_overlap_sums", "mvhdp_group_gamma_doc_statistics",
]

UNIQUE_ID_BYTES = 128


class MvhdpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mvhdp error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("num_topics", C.c_int32), ("num_modalities", C.c_int32),
                ("num_types", C.c_int32 * MAX_M), ("device", C.c_int32),
                ("doc_id_base", C.c_int64), ("flags", C.c_uint32)]


class HyperC(C.Structure):
    _fields_ = [("alpha", C.c_void_p),
                ("alpha_sum", C.c_double * MAX_M), ("beta", C.c_double * MAX_M),
                ("beta_sum", C.c_double * MAX_M), ("gamma", C.c_double * MAX_M),
                ("p_a", (C.c_double * MAX_M) * MAX_M), ("p_b", (C.c_double * MAX_M) * MAX_M),
                ("inactive", C.c_void_p)]


class SweepStatsC(C.Structure):
    _fields_ = [("tokens", C.c_int64), ("changed", C.c_int64), ("new_mass_cnt", C.c_int64),
                ("topic_doc_mass_cnt", C.c_int64), ("word_ftree_mass_cnt", C.c_int64),
                ("oov_skipped", C.c_int64), ("aborted_docs", C.c_int64), ("exact_fallbacks", C.c_int64),
                ("activated_topic", C.c_int32), ("activated_modality", C.c_int32),
                ("activation_key", C.c_int64), ("sweep_kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("activations", C.c_int32), ("reserved", C.c_int32)]


class DebugC(C.Structure):
    _fields_ = [("tok_dbg", C.c_void_p * MAX_M), ("n_trace", C.c_int32),
                ("trace_doc", C.c_void_p), ("trace_view", C.c_void_p), ("trace_pos", C.c_void_p),
                ("trace_out", C.c_void_p)]


class TuningC(C.Structure):
    _fields_ = [("force_primary", C.c_int32), ("narrow", C.c_int32), ("walk_fixed", C.c_int32), ("single_stream", C.c_int32),
                ("live16", C.c_int32), ("single_wave", C.c_int32),
                ("walk_theta", C.c_double * MAX_M), ("primary_min_share", C.c_double),
                ("learnt_walk_step", C.c_int32 * 4), ("tree_branch_share", C.c_double * MAX_M),
                ("live_overlap", C.c_int32), ("live_rows", C.c_int32)]


class GroupInfoC(C.Structure):
    _fields_ = [("local_members", C.c_int32), ("local_devices", C.c_int32), ("ranks", C.c_int32), ("first_rank", C.c_int32),
                ("rccl", C.c_int32), ("rccl_version", C.c_int32), ("exchange_chunks", C.c_int32), ("exchange_packed", C.c_int32),
                ("last_exchange_ms", C.c_double), ("last_exchange_bytes", C.c_int64)]


class PlanInputC(C.Structure):
    _fields_ = [("num_topics", C.c_int32), ("num_modalities", C.c_int32), ("num_entities", C.c_int64),
                ("max_entity_tokens", C.c_int64), ("entities_longer_than", C.c_int64 * 5),
                ("tokens_by_list_rounds", C.c_uint64 * 17), ("entities_by_class", C.c_uint64 * 8),
                ("flags", C.c_uint32), ("debug", C.c_int32), ("batch", C.c_int32), ("trees_current", C.c_int32),
                ("num_cus", C.c_int32), ("kernel_registers", (C.c_int32 * 3) * 6), ("inactive_topics", C.c_int32)]


class PlanOutputC(C.Structure):
    _fields_ = [("status", C.c_int32), ("segments", C.c_int32), ("primary_class", C.c_int32), ("register_resident", C.c_int32),
                ("need_full_trees", C.c_int32), ("dominant_class", C.c_int32), ("routed_prefix", C.c_int64),
                ("class_used", C.c_int32 * 6), ("class_map", C.c_int32 * 6), ("class_stream", C.c_int32 * 6),
                ("class_grid", C.c_int32 * 6), ("class_walk", C.c_int32 * 6), ("class_narrow", C.c_int32 * 6),
                ("class_register_resident", C.c_int32 * 6), ("class_lds_bytes", C.c_int64 * 6), ("class_theta0", C.c_double * 6),
                ("delta16", C.c_int32), ("live_rows", C.c_int32)]


_lib = None
_preloaded = []            # keeps the handles of the runtime libraries loaded on the library's behalf alive


def _torch_lib_dir():
    """Directory of the ROCm runtime libraries a PyTorch-ROCm wheel bundles (None when torch is not installed).
    Found from the package's location alone: torch is NOT imported here."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.submodule_search_locations:
        return None
    d = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    return d if os.path.isdir(d) else None


def mapped_runtime_libraries():
    """{'libamdhip64': [paths], 'libhsa-runtime64': [paths], 'librccl': [paths]} from /proc/self/maps: the distinct
    files of each ROCm runtime library mapped into this process."""
    found = {"libamdhip64": set(), "libhsa-runtime64": set(), "librccl": set()}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1] if "/" in line else ""
                base = os.path.basename(path)
                for name in found:
                    if base.startswith(name + ".so"):
                        found[name].add(os.path.realpath(path))
    except OSError:
        pass
    return {k: sorted(v) for k, v in found.items()}


def _preload_one_hip_runtime():
    """One HIP runtime per process, whatever is imported first.

    libmvhdp.so needs `libamdhip64.so.7` (RUNPATH /opt/rocm/lib); a PyTorch-ROCm wheel bundles its own copy with the
    SAME soname but asks for it by the file name `libamdhip64.so`, which the loader does not match against an already
    mapped `libamdhip64.so.7` from /opt/rocm: a process that creates a sampler first and touches torch.cuda later ends up
    with two HIP runtimes over two ROCr copies, and torch then fails with "No HIP GPUs are available".  The other order
    is fine (libmvhdp's `libamdhip64.so.7` matches the soname of torch's copy).  So when a torch wheel with a bundled
    runtime is installed, its libamdhip64.so is mapped FIRST,
    process-wide, and both libmvhdp.so and a later `import torch` resolve to that one copy.  A host without torch (the
    Java host of INTEGRATION.md) is not affected: the library then runs on /opt/rocm's runtime."""
    d = _torch_lib_dir()
    if d is None:
        return False
    already = mapped_runtime_libraries()
    if already["libamdhip64"] or already["libhsa-runtime64"]:
        # a HIP runtime is mapped already (torch imported first -- fine, libmvhdp.so will use that copy -- or a profiler's
        # preloaded tool library brought /opt/rocm's): nothing to choose any more
        return False
    path = os.path.join(d, "libamdhip64.so")
    if not os.path.exists(path):
        return False
    _preloaded.append(C.CDLL(path, mode=C.RTLD_GLOBAL))
    # RCCL is opened by the library itself, lazily, in mvhdp_group_create (dlopen by soname: a copy torch has mapped
    # already is re-used).  When torch is installed but not imported yet, point the library at the wheel's copy so that
    # a later `import torch` maps the same file (mapping it here, process-wide and ahead of torch, aborts at exit).
    rccl = os.path.join(d, "librccl.so")
    if os.path.exists(rccl):
        os.environ.setdefault("MVHDP_RCCL_LIB", rccl)
    return True


def load_library():
    """Load libmvhdp.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C mvtopicmodel_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    chose = _preload_one_hip_runtime()
    L = C.CDLL(LIB_PATH)
    maps = mapped_runtime_libraries()
    for name in ("libamdhip64", "libhsa-runtime64"):
        if chose and len(maps[name]) > 1:
            raise ImportError(
                f"two copies of {name} are mapped into this process ({', '.join(maps[name])}): a HIP runtime was loaded "
                "before mvtopicmodel_amd could pick one.  Import mvtopicmodel_amd (or torch) before any other library "
                "that links /opt/rocm's libamdhip64 in a process that also uses torch.")
    vp, i32, i64, u32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
    L.mvhdp_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.mvhdp_destroy.argtypes = [vp]
    L.mvhdp_last_error.argtypes = [vp]; L.mvhdp_last_error.restype = C.c_char_p
    L.mvhdp_version.restype = C.c_char_p
    L.mvhdp_set_corpus.argtypes = [vp, i32, i64, vp, vp]
    L.mvhdp_set_assignments.argtypes = [vp, i32, vp]
    L.mvhdp_set_view_presence.argtypes = [vp, i32, vp]
    L.mvhdp_get_assignments.argtypes = [vp, i32, vp]
    L.mvhdp_set_hyper.argtypes = [vp, C.POINTER(HyperC)]
    L.mvhdp_get_alpha.argtypes = [vp, vp, vp]
    L.mvhdp_build_counts.argtypes = [vp]
    L.mvhdp_build_trees.argtypes = [vp]
    L.mvhdp_build_inference_trees.argtypes = [vp]
    L.mvhdp_init_assignments_from_trees.argtypes = [vp, u64]
    L.mvhdp_get_counts.argtypes = [vp, i32, vp, vp]
    L.mvhdp_set_counts.argtypes = [vp, i32, vp, vp]
    L.mvhdp_get_tree.argtypes = [vp, i32, i32, vp]
    L.mvhdp_get_doc_topic_hist.argtypes = [vp, i32, vp, i32, vp, i32]
    L.mvhdp_get_count_histogram.argtypes = [vp, i32, vp, i32]
    L.mvhdp_view_overlap_sums.argtypes = [vp, vp]
    L.mvhdp_model_log_likelihood.argtypes = [vp, vp]
    L.mvhdp_doc_topic_proportions.argtypes = [vp, vp, i64, i64, vp]
    L.mvhdp_gamma_doc_statistics.argtypes = [vp, i32, C.c_double, u64, u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mvhdp_dp_table_statistics.argtypes = [vp, i32, vp, i32, vp, u64, u32, vp, vp]
    L.mvhdp_antoniak_draws.argtypes = [vp, i32, vp, vp, u64, u32, vp]
    L.mvhdp_sweep.argtypes = [vp, u32, u64, u32, vp, C.POINTER(DebugC), C.POINTER(SweepStatsC)]
    L.mvhdp_sweep_many.argtypes = [vp, u32, i32, u64, u32, vp]
    L.mvhdp_get_tuning.argtypes = [vp, C.POINTER(TuningC)]
    L.mvhdp_set_tuning.argtypes = [vp, C.POINTER(TuningC)]
    L.mvhdp_plan_probe.argtypes = [C.POINTER(PlanInputC), C.POINTER(TuningC), C.POINTER(PlanOutputC)]
    L.mvhdp_tuner_probe.argtypes = [i32, vp, vp, vp, i32, i32, vp]
    L.mvhdp_apply_delta.argtypes = [vp, i32, i32]
    L.mvhdp_apply_delta_begin.argtypes = [vp]
    L.mvhdp_apply_delta_rows.argtypes = [vp, i64, i64]
    L.mvhdp_apply_delta_end.argtypes = [vp, i32, i32]
    L.mvhdp_trees_current.argtypes = [vp]
    L.mvhdp_get_view_weights.argtypes = [vp, vp]
    L.mvhdp_device_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.mvhdp_counts_written.argtypes = [vp]
    L.mvhdp_set_stream.argtypes = [vp, vp]
    L.mvhdp_synchronize.argtypes = [vp]
    L.mvhdp_group_create.argtypes = [i32, C.POINTER(vp), C.POINTER(vp)]
    L.mvhdp_group_unique_id.argtypes = [vp]
    L.mvhdp_group_create_rank.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    L.mvhdp_group_destroy.argtypes = [vp]
    L.mvhdp_group_last_error.argtypes = [vp]; L.mvhdp_group_last_error.restype = C.c_char_p
    L.mvhdp_group_get_info.argtypes = [vp, C.POINTER(GroupInfoC)]
    L.mvhdp_group_set_exchange_chunks.argtypes = [vp, i32]
    L.mvhdp_group_build_counts.argtypes = [vp]
    L.mvhdp_group_sweep.argtypes = [vp, u32, u64, u32, vp]
    L.mvhdp_group_abort.argtypes = [vp]
    L.mvhdp_group_drain.argtypes = [vp]
    L.mvhdp_group_set_hyper.argtypes = [vp, C.POINTER(HyperC)]
    L.mvhdp_group_log_likelihood.argtypes = [vp, vp]
    L.mvhdp_group_doc_topic_hist.argtypes = [vp, i32, vp, i32, vp, i32]
    L.mvhdp_group_count_histogram.argtypes = [vp, i32, vp, i32]
    L.mvhdp_group_view_overlap_sums.argtypes = [vp, vp]
    L.mvhdp_group_gamma_doc_statistics.argtypes = [vp, i32, C.c_double, u64, u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    for name in ABI_SYMBOLS:
        f = getattr(L, name)  # raises AttributeError if the symbol is not exported
        if name not in ("mvhdp_last_error", "mvhdp_version", "mvhdp_group_last_error"):
            f.restype = C.c_int
    _lib = L
    return L
