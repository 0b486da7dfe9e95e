"""ctypes binding of include/ge_step.h.  Loads the in-tree libge_step.so and nothing else:
if the HIP library is missing this raises — the product never falls back to a CPU path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GE_LIB_PATH: another build of the same library (A/B runs, tools/abn.sh): the product .so is never overwritten
LIB_PATH = os.environ.get("GE_LIB_PATH") or os.path.join(_HERE, "libge_step.so")

GE_MAX_PHASES, GE_MAX_SEGMENTS, GE_NAME_LEN, GE_MAX_SLOTS = 32, 4, 64, 12
GE_ABI_VERSION = 5


class Literal(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("neg", C.c_uint8), ("num_field", C.c_uint8), ("pad", C.c_uint8),
                ("bases", C.c_uint16), ("lo", C.c_uint8), ("hi", C.c_uint8)]


class PhaseRow(C.Structure):
    _fields_ = [("phase_id", C.c_int32), ("completion", C.c_uint8), ("act", C.c_uint8),
                ("effect", C.c_uint8), ("n_terms", C.c_uint8), ("term_base", C.c_uint8 * 4),
                ("term_neg", C.c_uint8 * 4), ("n_branches", C.c_uint8), ("br_res", C.c_uint8 * 4),
                ("br_target", C.c_uint8 * 4), ("pad", C.c_uint8 * 3), ("name", C.c_char * GE_NAME_LEN),
                ("generic", C.c_uint8), ("n_clauses", C.c_uint8), ("clause_len", C.c_uint8 * 4), ("pad2", C.c_uint8 * 2),
                ("clause", (Literal * 4) * 4)]


class Table(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("pack", C.c_int32), ("n_phases", C.c_int32),
                ("rounds", C.c_int32), ("min_players", C.c_int32), ("init_fields", C.c_uint8 * 12),
                ("role_names", (C.c_char * GE_NAME_LEN) * 5), ("field_names", (C.c_char * GE_NAME_LEN) * GE_MAX_SLOTS),
                ("rows", PhaseRow * GE_MAX_PHASES)]


class SegmentDesc(C.Structure):
    _fields_ = [("table", C.POINTER(Table)), ("n_players", C.c_uint32), ("human_mask", C.c_uint32),
                ("n_rooms", C.c_uint64)]


class BatchDesc(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_room", C.c_uint64), ("n_segments", C.c_uint32),
                ("flags", C.c_uint32), ("device", C.c_int32), ("max_fuse", C.c_uint32),
                ("seg", SegmentDesc * GE_MAX_SEGMENTS)]


class Summary(C.Structure):
    _fields_ = [("rooms", C.c_uint64), ("finished", C.c_uint64), ("village_wins", C.c_uint64),
                ("wolf_wins", C.c_uint64), ("alive_players", C.c_uint64), ("sum_end_turn", C.c_uint64),
                ("end_turn_hist", C.c_uint64 * 16), ("score_hist", C.c_uint64 * 16),
                ("checksum", C.c_uint64), ("turn", C.c_uint64), ("games_recycled", C.c_uint64)]


SUMMARY_WORDS = C.sizeof(Summary) // 8

# every symbol include/ge_step.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = ["ge_table_compile_json", "ge_batch_create", "ge_batch_step", "ge_batch_reset", "ge_batch_set_turn", "ge_batch_inject_actions", "ge_batch_sync", "ge_batch_turn",
           "ge_batch_n_rooms", "ge_batch_read_rooms", "ge_batch_write_rooms", "ge_batch_read_events", "ge_batch_inject_action", "ge_batch_summary",
           "ge_batch_state", "ge_batch_set_timing", "ge_batch_kernel_time", "ge_batch_destroy",
           "ge_group_partition", "ge_batch_create_shard", "ge_group_create", "ge_group_size", "ge_group_shard", "ge_group_step", "ge_group_sync", "ge_group_summary", "ge_group_destroy",
           "ge_strerror", "ge_last_hip_error", "ge_last_rejected_room", "ge_last_comm_error", "ge_abi_version", "ge_device_count"]

_lib = None


def kernel_source_hash() -> str:
    """sha256 over the device code libge_step.so is built from - the kernels (ge_kernels.inl), the turn (ge_device.h), the record
    layout (ge_layout.h) - and the Makefile and peephole.sed, whose compiler flags and rewrites shape them: what ties a committed counter profile
    (profiles/pmc_*.json, tools/pmc_summary.py) to the kernels it was measured on (bench.py does not quote a profile of other
    kernels).  Host-side sources (ge_step.hip, ge_group.inl, ge_table.cpp) are not part of it."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    for name in ("Makefile", "peephole.sed", "ge_device.h", "ge_kernels.inl", "ge_layout.h"):
        with open(os.path.join(src, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # a fresh checkout (built artefacts are not in git): compile once if hipcc is here
        import shutil
        import subprocess
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-s"], check=False)
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C game_engine_amd/csrc` "
            "(or __graft_entry__.build()).  game_engine_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    lib.ge_table_compile_json.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(Table), C.c_char_p, C.c_size_t]
    lib.ge_batch_create.argtypes = [C.POINTER(BatchDesc), C.POINTER(vp)]
    lib.ge_batch_step.argtypes = [vp, u32, vp]
    lib.ge_batch_sync.argtypes = [vp]
    lib.ge_batch_reset.argtypes = [vp]
    lib.ge_batch_turn.argtypes = [vp, C.POINTER(u64)]
    lib.ge_batch_n_rooms.argtypes = [vp, C.POINTER(u64)]
    lib.ge_batch_read_rooms.argtypes = [vp, u64, u64, vp, C.c_size_t]
    lib.ge_batch_write_rooms.argtypes = [vp, u64, u64, vp]
    lib.ge_batch_read_events.argtypes = [vp, u64, u64, C.POINTER(u32), vp, C.c_size_t]
    lib.ge_batch_inject_action.argtypes = [vp, u64, u32, u32]
    lib.ge_batch_inject_actions.argtypes = [vp, u64, vp, vp, vp, vp]
    lib.ge_batch_set_turn.argtypes = [vp, u64]
    lib.ge_batch_summary.argtypes = [vp, C.POINTER(Summary)]
    lib.ge_batch_state.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(u32)]
    lib.ge_batch_set_timing.argtypes = [vp, C.c_int]
    lib.ge_batch_kernel_time.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(u64)]
    lib.ge_batch_destroy.argtypes = [vp]
    lib.ge_batch_destroy.restype = None
    lib.ge_strerror.argtypes = [C.c_int]
    lib.ge_strerror.restype = C.c_char_p
    if hasattr(lib, "ge_last_rejected_room"):
        lib.ge_last_rejected_room.restype = C.c_uint64
    # (GE_LIB_ANY_ABI: timing an older build through tools/abn.sh - only the entry points both versions share are used there)
    any_abi = bool(os.environ.get("GE_LIB_PATH") and os.environ.get("GE_LIB_ANY_ABI"))
    if lib.ge_abi_version() != GE_ABI_VERSION and not any_abi:
        raise ImportError("libge_step.so ABI version mismatch; rebuild it")
    if hasattr(lib, "ge_group_create") or not any_abi:
        lib.ge_group_create.argtypes = [C.POINTER(BatchDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        lib.ge_group_size.argtypes = [vp]
        lib.ge_group_shard.argtypes = [vp, C.c_int, C.POINTER(vp)]
        lib.ge_group_step.argtypes = [vp, u32]
        lib.ge_group_sync.argtypes = [vp]
        lib.ge_group_summary.argtypes = [vp, C.POINTER(Summary)]
        lib.ge_group_destroy.argtypes = [vp]
        lib.ge_group_destroy.restype = None
    if hasattr(lib, "ge_group_partition") or not any_abi:
        lib.ge_group_partition.argtypes = [C.POINTER(BatchDesc), C.c_int, C.c_int, C.POINTER(BatchDesc), C.POINTER(u64)]
        lib.ge_batch_create_shard.argtypes = [C.POINTER(BatchDesc), C.POINTER(u64), C.POINTER(vp)]
    _lib = lib
    return lib
