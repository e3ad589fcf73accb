"""CPU tests of the TacView writer (neuralplane_amd/envs/utils/acmi.py) against what the reference's render wrote
(tests/golden/acmi_kat.npz, tools/gen_golden.py gen_acmi)."""
import numpy as np

from neuralplane_amd.envs.utils import acmi


def test_enu_to_geodetic_matches_reference_function(golden_dir):
    d = np.load(f'{golden_dir}/acmi_kat.npz')
    lat, lon, h = acmi.enu_to_geodetic(d['enu'][:, 0], d['enu'][:, 1], d['enu'][:, 2])
    ref = d['geodetic']
    assert np.allclose(lat, ref[:, 0], rtol=0, atol=1e-10) and np.allclose(lon, ref[:, 1], rtol=0, atol=1e-10)
    assert np.allclose(h, ref[:, 2], rtol=0, atol=1e-6)      # the closed form loses ~1e-7 m at these heights in fp64
    # a non-trivial reference point: round trip through the inverse rotation
    x, y, z = acmi.enu_to_ecef([1200.0], [-3400.0], [560.0], 37.0, -122.0, 30.0)
    la, lo, hh = acmi.ecef_to_geodetic(x, y, z)
    assert abs(la[0] - 37.0) < 0.05 and abs(lo[0] + 122.0) < 0.05 and abs(hh[0] - 590.0) < 2.0


def test_recording_has_the_reference_layout_and_values(tmp_path, golden_dir):
    d = np.load(f'{golden_dir}/acmi_kat.npz')
    ref_header, ref_frames = acmi.parse_acmi(str(d['text']))
    rec = acmi.AcmiRecorder(str(tmp_path / 'tracks' / 'F16SimRecording-0.txt.acmi'))
    for row in d['states']:
        rec.frame(row[12] * 0.02, row[:12][None])
    header, frames = acmi.parse_acmi(open(rec.path).read())
    assert header == ref_header and len(frames) == len(ref_frames) == 4
    for (t, objs), (rt, robjs) in zip(frames, ref_frames):
        assert t == rt and len(objs) == len(robjs) == 1
        (oid, nums, props), (roid, rnums, rprops) = objs[0], robjs[0]
        assert oid == roid == 100 and props == rprops == {'Name': 'F16', 'Color': 'Red'}
        # the reference evaluates the geodetic conversion on float32 numpy scalars (+-0.5 m of ECEF round-off, it even
        # overflows an intermediate); this writer uses fp64 -> agree to 1e-5 deg (~1 m) and 1 m of altitude
        assert abs(nums[0] - rnums[0]) < 1e-5 and abs(nums[1] - rnums[1]) < 1e-5 and abs(nums[2] - rnums[2]) < 1.0
        assert np.allclose(nums[3:], rnums[3:], rtol=1e-6, atol=1e-9)


def test_two_colour_frame_for_an_engagement(tmp_path):
    rec = acmi.AcmiRecorder(str(tmp_path / 'c.acmi'))
    rec.frame(0.1, np.zeros((2, 12)), colors=('Red', 'Blue'))
    _, frames = acmi.parse_acmi(open(rec.path).read())
    assert [(o[0], o[2]['Color']) for o in frames[0][1]] == [(100, 'Red'), (101, 'Blue')]
