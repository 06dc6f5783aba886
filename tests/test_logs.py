"""SaveScalar text log and the vorticity picture (rlfluidcontrol_amd/logs.py) -- CPU checks of the formats."""
import struct
import zlib

import numpy as np

from rlfluidcontrol_amd import logs


def test_savescalar_format(tmp_path):
    p = tmp_path / "1.txt"
    lg = logs.SaveScalarLog(str(p))
    probes = np.linspace(-0.5, 0.5, 32).astype(np.float32)
    lg.add_data03(np.float32(0.0075), np.float32(0.5), np.float32(-1.0), (np.float32(9.069063), np.float32(8.5e-4)), probes)
    lg.add_data03(np.float32(0.015), 0.0, 0.0, (1.0, -2.0), probes)
    lg.finish()
    lines = p.read_text().split("\n")
    # SaveScalar.pde:28-33: two comment lines, each followed by an empty line
    assert lines[0].startswith("%% Force and pressure coefficients") and lines[1] == "" and lines[3] == ""
    assert lines[2] == "%% Fellowing: t, force.x, force.y"
    first = lines[4]
    assert first.endswith(" ") and first.startswith("0.0075 9.069063 8.5E-4 0.5 -1.0 -0.5 ")  # Java float text
    toks = first.split()
    assert len(toks) == 5 + 32 and toks[-1] == "0.5"
    assert [np.float32(t) for t in toks[5:]] == list(probes)  # shortest round-trip decimals parse back exactly
    assert lines[5].startswith("0.015 1.0 -2.0 0.0 0.0 ")


def test_flood_colours_and_png(tmp_path):
    c = logs.flood_colors(np.array([-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 3.0]))
    assert c[0].tolist() == [0, 0, 255] and c[1].tolist() == [0, 0, 255]  # saturated blue at/below the low end
    assert c[3].tolist() == [255, 255, 255]                                # white at mid-range
    assert c[5].tolist() == [255, 0, 0] and c[6].tolist() == [255, 0, 0]   # saturated red at/above the high end
    assert c[2].tolist() == [128, 128, 255] and c[4].tolist() == [255, 128, 128]
    vort = np.zeros((10, 6), np.float32)
    vort[2, 1], vort[7, 4] = 1.0, -1.0
    h, w = logs.vorticity_png(str(tmp_path / "v.png"), vort, zoom=2)
    assert (h, w) == (12, 20)
    data = (tmp_path / "v.png").read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    ww, hh, depth, ctype = struct.unpack(">IIBB", data[16:26])
    assert (ww, hh, depth, ctype) == (20, 12, 8, 2)
    n = struct.unpack(">I", data[33:37])[0]
    assert data[37:41] == b"IDAT"
    raw = zlib.decompress(data[41:41 + n])
    rows = np.frombuffer(raw, np.uint8).reshape(12, 1 + 20 * 3)
    img = rows[:, 1:].reshape(12, 20, 3)
    assert img[2, 4].tolist() == [255, 0, 0] and img[8, 14].tolist() == [0, 0, 255] and img[0, 0].tolist() == [255, 255, 255]
