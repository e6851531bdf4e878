"""One-off soak: the randomized parity cases of tests/test_hip_driver.py and tests/test_hip_random_configs.py
with seeds the suite does not use.  usage: python tools/soak_random.py [first_seed] [count]"""
import os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytest  # noqa: F401
import test_hip_driver as td
import test_hip_random_configs as tr


class MP:  # the two monkeypatch methods the tests use
    def __init__(self): self.saved = {}
    def setenv(self, k, v):
        self.saved.setdefault(k, os.environ.get(k)); os.environ[k] = v
    def undo(self):
        for k, v in self.saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        self.saved = {}


os.environ.setdefault("ROADSURF_HIP_CLUSTER", "1")  # (as tests/conftest.py: the plan-order path also for small batches)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    mp = MP()
    try:
        td.test_random_driver_case_matches_checker(seed, mp)
        print("driver seed", seed, "ok", flush=True)
    except AssertionError as e:
        bad += 1
        print("driver seed", seed, "FAILED", str(e)[:300], flush=True)
    finally:
        mp.undo()
for seed in range(first, first + count // 2):
    try:
        tr.test_random_configuration_is_bit_identical(seed)
        print("config seed", seed, "ok", flush=True)
    except AssertionError as e:
        bad += 1
        print("config seed", seed, "FAILED", str(e)[:300], flush=True)
    except Exception as e:  # pytest.skip and friends
        print("config seed", seed, type(e).__name__, str(e)[:100], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
