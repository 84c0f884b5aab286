"""GPU: a few rounds of the randomised differential run (tests/fuzz_parity.py)."""
import os
import runpy

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_rounds(monkeypatch, capsys):
    monkeypatch.setenv("IIV_FUZZ_ROUNDS", "10")
    monkeypatch.setenv("IIV_FUZZ_SEED", "77")
    runpy.run_path(os.path.join(os.path.dirname(__file__), "fuzz_parity.py"), run_name="__main__")
    assert "all equal" in capsys.readouterr().out
