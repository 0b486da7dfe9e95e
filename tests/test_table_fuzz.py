"""Hardening of the host-side C ABI (CPU): ge_table_compile_json must never crash, whatever bytes
it is given — it either fills the table or returns GE_ERR_DSL / GE_ERR_ARG.  Run once against the
normal library and once against an AddressSanitizer + UBSan build of ge_table.cpp alone
(sanitizers are CPU-only on this pool)."""
import copy
import ctypes as C
import json
import os
import subprocess
import sys

import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from conftest import ROOT, load_dsl
from game_engine_amd import _lib

json_values = st.recursive(
    st.none() | st.booleans() | st.integers(-2**40, 2**40) | st.floats(allow_nan=False) | st.text(max_size=12),
    lambda ch: st.lists(ch, max_size=4) | st.dictionaries(st.text(max_size=8), ch, max_size=4), max_leaves=20)


def _compile(lib, data: bytes, rounds=1):
    t = _lib.Table()
    err = C.create_string_buffer(256)
    st_ = lib.ge_table_compile_json(data, len(data), rounds, C.byref(t), err, len(err))
    assert st_ in (0, -1, -2), st_
    assert b"\\0" not in err.raw[:1] or True
    return st_, t


@settings(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.binary(max_size=200))
def test_random_bytes_do_not_crash(data):
    _compile(_lib.load(), data)


@settings(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(json_values)
def test_random_json_documents_do_not_crash(doc):
    st_, _ = _compile(_lib.load(), json.dumps(doc).encode())
    assert st_ != 0 or isinstance(doc, dict)


@settings(max_examples=200, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.data())
def test_mutated_real_dsl_does_not_crash(data):
    """Random edits of the shipped Werewolf DSL: delete / replace random sub-trees."""
    d = copy.deepcopy(load_dsl("werewolf-(mafia)"))
    for _ in range(data.draw(st.integers(1, 4))):
        node = d
        for _depth in range(data.draw(st.integers(1, 5))):
            if isinstance(node, dict) and node:
                k = data.draw(st.sampled_from(sorted(node.keys(), key=str)))
            elif isinstance(node, list) and node:
                k = data.draw(st.integers(0, len(node) - 1))
            else:
                break
            if data.draw(st.booleans()) or not isinstance(node[k], (dict, list)):
                node[k] = data.draw(json_values)
                break
            node = node[k]
    st_, t = _compile(_lib.load(), json.dumps(d).encode())
    if st_ == 0:
        assert 1 <= t.n_phases <= 32
        for i in range(t.n_phases):
            r = t.rows[i]
            assert r.n_terms <= 4 and r.n_branches <= 4 and all(r.br_target[j] < t.n_phases for j in range(r.n_branches))


def test_truncated_real_dsl_every_prefix():
    text = json.dumps(load_dsl("two-truths-and-a-lie")).encode()
    lib = _lib.load()
    for cut in range(0, len(text), 97):
        assert _compile(lib, text[:cut])[0] in (-1, -2)
    assert _compile(lib, text)[0] == 0


def test_compiler_under_address_and_ub_sanitizers(tmp_path):
    """ge_table.cpp rebuilt alone with -fsanitize=address,undefined (no HIP in it) and driven by a
    tiny C harness over the shipped DSLs, truncations and garbage."""
    src = os.path.join(ROOT, "game_engine_amd", "csrc", "ge_table.cpp")
    harness = tmp_path / "h.cpp"
    harness.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "%s/include/ge_step.h"
int main(int argc, char **argv) {
    int ok = 0;
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb"); if (!f) return 2;
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        char *buf = (char *)malloc(n + 1); if (fread(buf, 1, n, f) != (size_t)n) return 3; fclose(f);
        ge_game_table t; char err[128];
        if (ge_table_compile_json(buf, n, 1, &t, err, sizeof err) == 0) ok++;
        for (long cut = 0; cut < n; cut += 211) { char *c = (char *)malloc(cut ? cut : 1); memcpy(c, buf, cut); ge_table_compile_json(c, cut, 1, &t, err, sizeof err); free(c); }
        for (long i = 0; i < n; i += 53) { char s = buf[i]; buf[i] = (char)(i * 7); ge_table_compile_json(buf, n, 1, &t, err, 8); buf[i] = s; }
        free(buf);
    }
    printf("compiled %%d\n", ok);
    return ok == argc - 1 ? 0 : 1;
}''' % ROOT)
    exe = tmp_path / "h"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-o", str(exe), str(harness), src])
    dsls = [os.path.join(ROOT, "tests", "golden", "dsl", f) for f in ("werewolf-(mafia).json", "two-truths-and-a-lie.json")]
    out = subprocess.run([str(exe), *dsls], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "compiled 2" in out.stdout
