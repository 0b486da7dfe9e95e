"""Host-only code of libge_step.so under AddressSanitizer + UBSan (CPU; GPU sanitizers are not available on this pool):
ge_table.cpp (the DSL compiler) and csrc/ge_host.h (table rows, the generic rows' literal image, restart template, room
view <-> packed record conversion, the checks of ge_batch_write_rooms, the device group's sharding arithmetic group_partition) built with g++ -fsanitize=address,undefined and driven
by tests/native/host_sanitize.cpp over the shipped DSLs, the reference's draft and every grammar variant of the goldens, for
every player count and record layout.  (ge_table.cpp alone against garbage input: tests/test_table_fuzz.py.)"""
import json
import os
import subprocess

from conftest import GOLD, ROOT, load_dsl
from oracle import dsl_variants


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    exe = tmp_path / "host_sanitize"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-o", str(exe),
                           os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"),
                           os.path.join(ROOT, "game_engine_amd", "csrc", "ge_table.cpp")])
    files = [os.path.join(GOLD, "dsl", f) for f in sorted(os.listdir(os.path.join(GOLD, "dsl"))) if f.endswith(".json")]
    for name, (game, builder, _rounds) in sorted(dsl_variants.VARIANTS.items()):
        p = tmp_path / f"variant_{name}.json"
        p.write_text(json.dumps(builder(load_dsl(game)), ensure_ascii=False), encoding="utf-8")
        files.append(str(p))
    out = subprocess.run([str(exe), *files], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
    assert f"compiled {2 * len(files)}" in out.stdout
