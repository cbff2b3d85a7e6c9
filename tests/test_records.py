import os

import numpy as np

from sbdart_amd.records import read_records, write_records

from conftest import GOLDEN


def test_roundtrip(tmp_path):
    recs = read_records(os.path.join(GOLDEN, "sbchk5.sbdrec"))
    p = tmp_path / "x.sbdrec"
    write_records(str(p), recs)
    again = read_records(str(p))
    assert len(again) == len(recs)
    for a, b in zip(recs, again):
        assert a.nstr == b.nstr and a.flags == b.flags and a.kd == b.kd and a.iwl == b.iwl
        for f in ("dtauc", "ssalb", "temper", "pmom", "umu", "phi", "rfldn", "uu"):
            assert np.array_equal(getattr(a, f), getattr(b, f))
    write_records(str(p), [r.inputs_only() for r in recs])
    assert not read_records(str(p))[0].has_out()
