"""A seeded slice of the randomised differential run (tools/fuzz_parity.py): random small architectures (both front ends,
grouped KV heads, head sizes 16-64, windows 3-72, 1-3 layers, three decoder widths), batch sizes, lengths, chunk plans and
push plans; forward, chunked_decode, the ragged decode wrapper and a stream against the CPU oracle, fp32 at the 1e-3 / 60 dB
bar and bf16 (conditioned weights) at 30 dB."""
import importlib.util
import os
import random

import pytest

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location(
    "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzz)


@pytest.mark.parametrize("seed", [11, 12])
def test_random_architectures_and_plans_against_the_oracle(seed, capsys):
    rng = random.Random(seed)
    for i in range(12):
        ok = fuzz.one_case(i, rng, verbose=False)
        assert ok, capsys.readouterr().out


def test_random_encoders_and_lengths_against_the_oracle(capsys):
    """The encode side the same way: random small encoders (strides, widths, heads, windows), batch sizes and audio lengths
    that no stride divides - fp32 embeddings within 1e-4 of the oracle's, codes equal up to near-ties."""
    rng = random.Random(21)
    for i in range(16):
        ok = fuzz.encode_case(i, rng, verbose=False)
        assert ok, capsys.readouterr().out
