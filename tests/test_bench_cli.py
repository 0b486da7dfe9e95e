"""bench.py's command line (the driver's contract): `--gpus N` must start N ranks itself when it is
not already one rank of a torch.distributed.run job, before anything touches the GPU, and a
`--gpus` / WORLD_SIZE mismatch must fail instead of silently running one rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_cli_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_n_spawns_n_ranks_before_touching_the_gpu(monkeypatch, capsys):
    bench = _bench_module()
    seen = {}

    class Done:
        returncode = 0
        stdout = b'NCCL version banner\n{"metric": "room-phase steps/sec", "n_gpus": 4}\n'

    def fake_run(cmd, env=None, stdout=None):
        seen["cmd"], seen["env"] = cmd, env
        seen["torch_loaded"] = "torch" in sys.modules and getattr(sys.modules["torch"], "cuda", None) is not None and \
            sys.modules["torch"].cuda.is_initialized()
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    bench.main()
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not seen["torch_loaded"]                        # the parent never initialised the GPU
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 4   # exactly rank 0's line is relayed


def test_world_size_mismatch_fails():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"WORLD_SIZE=2" in p.stderr and p.stdout == b""


def test_spawned_rank_failure_is_reported(monkeypatch):
    bench = _bench_module()

    class Failed:
        returncode = 3
        stdout = b"rank 1 died\n"

    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: Failed())
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3


@pytest.mark.gpu
def test_gpus_2_rehearsal_on_one_card_reports_two_ranks():
    """`python bench.py --gpus 2` end to end on the one-GPU test box: both ranks share GPU 0 and
    the collectives run over gloo (GE_DIST_BACKEND=gloo); the line must say n_gpus 2 and count
    both shards' rooms."""
    env = dict(os.environ, GE_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = p.stdout.decode().strip().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["summary"]["rooms"] == 2 * 65536
    assert d["config"]["room_phase_steps_per_bench_step"] == 2 * 65536 * 1024
    assert d["value"] > 1e9
    # a multi-GPU record is self-sufficient: rank 0's CPU row, the roofline block, and per workload a roofline of its own
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 1e5 and "sample" in cb
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "bound_actual"):
        assert k in d["roofline"], k
    # one --gpus N command also yields BASELINE configs[3] (C4) and configs[4] (C5): every rank's share, timed all-gather
    ow = d["other_workloads"]
    assert set(ow) == {"c4", "c5"}
    for key, rooms_per_gpu in (("c4", 2097152), ("c5", 1048576)):
        w = ow[key]
        for k in ("rooms_total", "value", "algorithmic_frac", "summary_allgather_ms", "checksum", "turns_stepped", "ms_per_launch"):
            assert k in w, (key, k)
        assert w["rooms_total"] == 2 * rooms_per_gpu == w["summary"]["rooms"] and w["value"] > 1e9
        rf = w["roofline"]
        assert rf["bound"] == "hbm" and rf["algorithmic"] is True and rf["bound_actual"] == "valu-issue" and rf["peak"] == 2 * 8000.0
        assert abs(rf["frac"] - w["algorithmic_frac"]) < 1e-12 and "issue" in rf
    # C4, the configuration BASELINE names for 8 GPUs: every rank's share through single-turn launches, and a CPU row
    hs = ow["c4"]["hbm_streaming"]
    assert hs["rooms"] == 2097152 and hs["bytes_per_launch"] == 2 * 40 * 2097152 and 0 < hs["frac_min_over_ranks"] <= hs["frac"] * 1.0001
    assert hs["fits_infinity_cache"] is True and "memory-side" in hs["what_frac_is"]
    assert ow["c4"]["cpu_baseline"]["kind"] == "port" and ow["c5"]["hbm_streaming"] is None
    # the C4 checksum of the two-rank job = the sum of the two shards stepped on their own (rooms keep their global index)
    from conftest import load_dsl
    from game_engine_amd import GameTable, RoomBatch
    c4 = ow["c4"]
    tb = GameTable(load_dsl("werewolf-(mafia)"))
    total = 0
    for r in range(2):
        with RoomBatch([(tb, 12, 2097152)], seed=c4["seed"], first_room=r * 2097152, max_fuse=c4["turns_fused_per_launch"], restart=True) as b:
            b.step(c4["turns_stepped"])
            total = (total + b.summary()["checksum"]) % (1 << 64)
    assert total == c4["checksum"]


@pytest.mark.gpu
def test_n1_line_carries_the_physical_figures_in_the_roofline_block():
    """The driver keeps `roofline` whole: the physical (single-turn launch) figures of the run live inside it, each with its parity
    check, and the line ends with `roofline` and `cpu_baseline` so that a tail of it still shows them (round-4 verdict, item 2)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--no-other-shapes", "--no-from-init"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert list(d)[-2:] == ["roofline", "cpu_baseline"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "physical", "fused", "bound_actual", "frac_of_actual_bound", "frac_of_priced_bound"):
        assert k in rf, k
    # the vector pipe's share at measured instruction prices: quoted only from a profiles/valu_mix.json made for this device code
    if rf["frac_of_priced_bound"] is not None:
        assert rf["frac_of_actual_bound"] < rf["frac_of_priced_bound"] < 1.05 and d["issue"]["valu_mean_price_cycles"] >= 2.0
    rows = rf["physical"]
    assert len(rows) == 1 and rows[0]["rooms"] == 65536 and rows[0]["parity"] is True and rows[0]["hbm"] is False
    assert 0.02 < rows[0]["frac"] < 1.0 and rows[0]["hbm_floor_frac"] == 0.0
    assert rf["fused"][0]["steps_per_s"] == d["value"] and d["cpu_baseline"]["kind"] == "port"
    assert "note" not in rf and len(lines[0]) < 8192             # without the other shapes the whole line fits a tail
