"""The committed golden vectors are what the oracle produces today (guards against silent
oracle drift).  CPU only."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import golden_util as G

IDX = G.index()


@pytest.mark.parametrize("name", sorted(IDX))
def test_oracle_reproduces_golden(oracle, name):
    e = IDX[name]
    u = oracle.Uniforms.from_buffer_copy(G.uniforms_bytes(e))
    img, cnt = oracle.render(u, tuple(e["limits"]), e["cmd_count"], G.words(e), e["W"], e["H"], threads=4,
                             want_counters=True, materials=e.get("materials"))
    assert hashlib.sha256(img.tobytes()).hexdigest() == e["sha256"]
    assert cnt == e["counters"]
    if "file" in e:
        assert G.load_image(e).tobytes() == img.tobytes()
