"""CPU: submission-file writers (MOT17 text, MOTS20 text with COCO compressed RLE)."""
import numpy as np

from trackformer_amd import results_io as rio


def test_mot_results_round_trip(tmp_path):
    results = {0: {0: {'bbox': np.array([10., 20., 50., 80.], np.float32), 'score': 0.9, 'obj_ind': 3},
                   2: {'bbox': np.array([12.5, 21., 52., 83.], np.float32), 'score': 0.8, 'obj_ind': 3}},
               4: {1: {'bbox': np.array([0., 0., 5., 6.], np.float32), 'score': 0.7, 'obj_ind': 9}}}
    path = tmp_path / "out" / "MOT17-02.txt"
    rio.write_mot_results(results, str(path))
    lines = path.read_text().strip().splitlines()
    # mot17_sequence.py:227-240: 1-based frame / id / coordinates, width = x2 - x1 + 1
    assert lines[0] == "1,1,11.0,21.0,41.0,61.0,-1,-1,-1,-1"
    assert lines[2].startswith("2,5,1.0,1.0,6.0,7.0")
    back = rio.load_mot_results(str(path))
    assert sorted(back) == [0, 4] and sorted(back[0]) == [0, 2]
    for tid in results:
        for f in results[tid]:
            np.testing.assert_allclose(back[tid][f]['bbox'], results[tid][f]['bbox'])


def test_rle_known_vectors():
    # hand-derived from cocoapi maskApi.c (rleToString): small counts are '0' + count, counts >= 16 use a
    # continuation character, counts after the third are stored as differences to counts[i-2]
    assert rio.rle_to_string([1, 3]) == b"13"
    assert rio.rle_to_string([5, 2, 7, 3]) == b"5271"               # 3 - counts[1] = 1
    assert rio.rle_to_string([100]) == b"T3"
    assert rio.rle_to_string([4, 4, 4, 1]) == b"444M"           # 1 - 4 = -3 -> 'M'
    m = np.array([[0, 1], [1, 1]], dtype=bool)                    # column-major: 0 1 1 1
    assert rio.rle_counts(m) == [1, 3]
    assert rio.rle_encode(m) == {'size': [2, 2], 'counts': b"13"}
    assert rio.rle_counts(np.ones((2, 3), bool)) == [0, 6]        # masks starting with 1: leading 0 run


def test_rle_round_trip_random_masks():
    rng = np.random.default_rng(0)
    for shape, p in [((37, 53), 0.5), ((480, 640), 0.02), ((5, 1), 0.9), ((1, 7), 0.3), ((64, 64), 0.0)]:
        mask = rng.random(shape) < p
        rle = rio.rle_encode(mask)
        assert rle['size'] == list(shape)
        assert rio.rle_from_string(rle['counts']) == rio.rle_counts(mask)
        np.testing.assert_array_equal(rio.rle_decode(rle), mask)
    blob = np.zeros((480, 640), bool)
    blob[100:300, 200:420] = True                                  # long runs: multi-character counts
    np.testing.assert_array_equal(rio.rle_decode(rio.rle_encode(blob)), blob)


def test_mots_results_file(tmp_path):
    mask = np.zeros((6, 8), bool)
    mask[2:4, 3:6] = True
    path = tmp_path / "MOTS20-02.txt"
    rio.write_mots_results({7: {4: {'mask': mask}}}, str(path))
    frame, tid, cls, h, w, counts = path.read_text().strip().split(' ')
    assert (frame, tid, cls, h, w) == ("5", "8", "2", "6", "8")
    np.testing.assert_array_equal(rio.rle_decode({'size': [6, 8], 'counts': counts.encode()}), mask)
