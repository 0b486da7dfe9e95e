"""The C-ABI library loads, exports every symbol include/ge_step.h declares, and refuses to run
without a GPU instead of falling back to a CPU path.  No compute calls."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from game_engine_amd import GameTable, GeError, RoomBatch, _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "ge_step.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ge_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert set(names) == set(_lib.SYMBOLS)
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.ge_abi_version() == _lib.GE_ABI_VERSION == 5


def test_struct_sizes_match_header():
    # sizes the N-API / cgo / ctypes bindings rely on
    from game_engine_amd.stepper import ROOM_VIEW_DTYPE
    assert ROOM_VIEW_DTYPE.itemsize == 20 + 16 * 12 + 16
    assert C.sizeof(_lib.Literal) == 8
    assert C.sizeof(_lib.PhaseRow) == 4 + 4 + 4 + 4 + 1 + 4 + 4 + 3 + 64 + 8 + 4 * 4 * 8
    assert C.sizeof(_lib.Summary) == 8 * 41


def test_strerror_and_argument_errors():
    lib = _lib.load()
    assert lib.ge_strerror(0) == b"ok"
    assert b"no HIP device" in lib.ge_strerror(-3)
    assert lib.ge_batch_create(None, None) == -1
    assert lib.ge_batch_step(None, 1, None) == -1
    assert lib.ge_table_compile_json(None, 0, 1, None, None, 0) == -1


def test_no_cpu_fallback(dsl_ww):
    """Without a GPU the product must fail loudly (GE_ERR_NO_DEVICE), never emulate."""
    lib = _lib.load()
    if lib.ge_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(GeError) as e:
        RoomBatch([(GameTable(dsl_ww), 8, 16)])
    assert e.value.status == -3


def test_device_group_refuses_without_gpu_and_bad_arguments(dsl_ww):
    """ge_group_* (single process, N devices, RCCL all-gather inside the library): argument errors, and no CPU path."""
    from game_engine_amd import RoomGroup
    lib = _lib.load()
    assert lib.ge_group_create(None, None, 0, None) == -1
    assert lib.ge_group_size(None) == -1 and lib.ge_group_step(None, 1) == -1 and lib.ge_group_summary(None, None) == -1
    assert b"RCCL" in lib.ge_strerror(-9)
    tb = GameTable(dsl_ww)
    if lib.ge_device_count() > 0:
        pytest.skip("a GPU is present: covered by tests/test_gpu_group.py")
    with pytest.raises(GeError) as e:
        RoomGroup([(tb, 8, 64)], devices=[0])
    assert e.value.status == -3                          # GE_ERR_NO_DEVICE: no CPU fallback here either


def test_inject_actions_on_a_dead_handle_raises():
    """A failure of ge_batch_inject_actions itself (here: no batch behind the handle) leaves the per-action status array
    untouched; the host must raise, not report every action as applied (0 = applied)."""
    import numpy as np
    b = RoomBatch.__new__(RoomBatch)                 # a host object whose batch is gone (as after close())
    b._lib, b._h = _lib.load(), None
    with pytest.raises(GeError) as e:
        b.inject_actions([0, 1], [1, 1], [2, 2])
    assert e.value.status == -1
    with pytest.raises(GeError):
        b.inject_actions(np.zeros(0, np.uint64), [], [])       # even an empty call fails on a dead handle


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under game_engine_amd/ may reference it."""
    pkg = os.path.join(ROOT, "game_engine_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".inl", ".cpp", ".h", ".js", ".ts", ".cc")):
                src = open(os.path.join(d, f), encoding="utf-8").read()
                for bad in ("import oracle", "from oracle", "libge_oracle", "ge_oracle", "orc_run", "refharness"):
                    assert bad not in src, (os.path.join(d, f), bad)


def test_header_is_plain_c_and_links(tmp_path):
    """include/ge_step.h is the boundary: it must compile as C99 (-pedantic) and a C program must link against the
    in-tree library and call it (no GPU needed for the table compiler and the version / error-string entry points)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc is not available")
    from conftest import ROOT
    src = tmp_path / "consumer.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "ge_step.h"
int main(void) {
    static ge_game_table t;
    char err[256];
    const char *bad = "{\"declaration\": {}, \"phases\": {}}";
    if (ge_abi_version() != GE_ABI_VERSION) return 2;
    if (ge_table_compile_json(bad, strlen(bad), 1, &t, err, sizeof err) != GE_ERR_DSL) return 3;
    if (!ge_strerror(GE_ERR_DSL) || !strlen(err)) return 4;
    printf("%d %s\n", ge_device_count(), err);
    return 0;
}
''')
    exe = tmp_path / "consumer"
    libdir = os.path.join(ROOT, "game_engine_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                           "-L", libdir, "-lge_step", "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert "needs top-level" in out.stdout


def test_group_partition_through_the_c_abi_needs_no_device():
    """ge_group_partition (the device group's sharding arithmetic, ABI 5) is host-only: parts tile every segment and carry the
    global index of their first room; the Python hosts' reassembly and RoomGroup read-back walk the same parts."""
    import ctypes as C
    from game_engine_amd import _lib
    from game_engine_amd.stepper import partition
    lib = _lib.load()
    for rooms, first_room in (([16777216], 0), ([8388608, 8388608], 1 << 40), ([1000003, 64, 777, 4099], 0xFFFFFFFF00)):
        d = _lib.BatchDesc()
        d.first_room, d.n_segments, d.seed = first_room, len(rooms), 5
        for k, r in enumerate(rooms):
            d.seg[k].n_rooms, d.seg[k].n_players = r, 8
        for n in (1, 2, 3, 8, 64):
            nxt, acc = [], first_room
            for r in rooms:
                nxt.append(acc)
                acc += r
            for i in range(n):
                sd, first = partition(d, n, i)
                assert first == nxt and sd.seed == 5 and sd.n_segments == len(rooms)
                for k, r in enumerate(rooms):
                    assert sd.seg[k].n_rooms == r * (i + 1) // n - r * i // n > 0
                    nxt[k] += sd.seg[k].n_rooms
            acc = first_room
            for k, r in enumerate(rooms):
                acc += r
                assert nxt[k] == acc
    sh, first = _lib.BatchDesc(), (C.c_uint64 * 4)()
    assert lib.ge_group_partition(C.byref(d), 65, 0, C.byref(sh), first) == -1          # a segment has only 64 rooms
    assert lib.ge_group_partition(C.byref(d), 2, 2, C.byref(sh), first) == -1
    assert lib.ge_group_partition(None, 2, 0, C.byref(sh), first) == -1
    assert lib.ge_batch_create_shard(C.byref(sh), None, None) == -1
    assert lib.ge_last_rejected_room() == 0xFFFFFFFFFFFFFFFF               # no write has been refused on this thread
