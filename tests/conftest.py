import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_dsl(game: str) -> dict:
    """The reference's YAML game DSL as committed JSON fixture (tests/golden/dsl, written by
    oracle/refharness/make_golden.py from yaml.safe_load of /root/reference/games/<game>.yaml)."""
    with open(os.path.join(GOLD, "dsl", f"{game}.json"), encoding="utf-8") as f:
        return json.load(f)


def golden_dsl(g: dict) -> dict:
    """The DSL a golden file was produced with: the game's fixture, or a grammar variant of it
    (oracle/dsl_variants.py; the variants are functions of the committed base DSL)."""
    dsl = load_dsl(g["game"])
    if g.get("variant"):
        from oracle import dsl_variants
        dsl = dsl_variants.build(g["variant"], dsl)
    return dsl


def golden_files():
    return sorted(f for f in os.listdir(GOLD) if f.startswith("traj_") and f.endswith(".json"))


def restart_files():
    return sorted(f for f in os.listdir(GOLD) if f.startswith("restart_") and f.endswith(".json"))


def human_files():
    return sorted(f for f in os.listdir(GOLD) if f.startswith("human_") and f.endswith(".json"))


def load_golden(name: str) -> dict:
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def dsl_ww():
    return load_dsl("werewolf-(mafia)")


@pytest.fixture(scope="session")
def dsl_tt():
    return load_dsl("two-truths-and-a-lie")
