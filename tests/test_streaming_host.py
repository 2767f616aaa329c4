"""Window index logic of the streaming path (host side, CPU): known answers restating lib/_img_utils.py:42-55 and
lib/utils/_dataset_demo.py:91-96 (the reference module itself needs cv2/skimage/torchvision, absent here)."""
import numpy as np

from pmce_amd.streaming import demo_window_list, window_indices


def brute(num_frames, seqlen=16, stride=1):
    idx = np.arange(num_frames)
    chunks = [idx[i:i + seqlen] for i in range(0, num_frames - seqlen + 1, stride)]
    sf = [[c[0], c[-1]] for c in chunks]
    vibe = [idx[i:i + 16] for i in range(0, num_frames - 16 + 1, 16)]
    if stride != seqlen:
        for j in range(1, len(sf) + 1):
            if sf[-j][-1] == vibe[-1][-1]:
                if j != 1:
                    sf = sf[:-j + 1]
                break
    return np.array(sf)


def test_window_indices_match_bruteforce():
    for L in (16, 17, 31, 32, 33, 47, 48, 100, 1000):
        assert np.array_equal(window_indices(L), brute(L)), L
    assert window_indices(15).shape == (0, 2)
    w = window_indices(40)                       # last full VIBE chunk ends at frame 31 -> windows 0..16
    assert w[0].tolist() == [0, 15] and w[-1].tolist() == [16, 31] and len(w) == 17
    assert len(window_indices(32, stride=16)) == 2


def test_demo_window_list():
    L = 40
    w = demo_window_list(L)
    assert len(w) == L                                      # one prediction per frame
    assert w[:8].tolist() == [[i, i] for i in range(8)]
    assert w[8].tolist() == [0, 15] and w[8 + (L - 16)].tolist() == [L - 16, L - 1]
    assert w[-7:].tolist() == [[L - 7 + i, L - 7 + i] for i in range(7)]
