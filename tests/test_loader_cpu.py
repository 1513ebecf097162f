"""Loader: chunking, overlap, decode equality -- host arrays, no GPU."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_gray
from visual_slam_amd import loader


def test_paths_and_decode_match_the_fixtures():
    paths = loader.load_image_paths(os.path.join(GOLDEN, "kitti02"))
    assert [os.path.basename(p) for p in paths] == ["%06d.png" % i for i in range(4)]
    for p in paths:
        assert np.array_equal(loader.decode_gray(p), load_gray(p))


@pytest.mark.parametrize("chunk,overlap,want", [(2, 0, [(0, 2), (2, 4)]), (3, 1, [(0, 3), (2, 4)]), (2, 1, [(0, 2), (1, 3), (2, 4)]), (8, 1, [(0, 4)])])
def test_chunks_cover_every_consecutive_pair(chunk, overlap, want):
    paths = loader.load_image_paths(os.path.join(GOLDEN, "kitti02"))
    ld = loader.FrameLoader(paths, chunk=chunk, overlap=overlap, workers=2)
    assert ld.chunks == want
    got = list(ld)
    assert len(got) == len(want)
    full = np.stack([load_gray(p) for p in paths])
    for (a, b), arr in zip(want, got):
        assert arr.dtype == np.uint8 and np.array_equal(arr, full[a:b])


def test_empty_and_bad_arguments():
    assert list(loader.FrameLoader([], chunk=4)) == []
    with pytest.raises(ValueError):
        loader.FrameLoader(["a"], chunk=1, overlap=1)


def test_process_pool_decoding_matches_threads():
    """processes=N: spawned worker processes decode into a shared-memory ring (threads stop scaling at the GIL)."""
    paths = loader.load_image_paths(os.path.join(GOLDEN, "kitti02")) * 3       # 12 frames, 5 chunks of 3 with overlap 1
    want = list(loader.FrameLoader(paths, chunk=3, overlap=1, workers=2))
    got = list(loader.FrameLoader(paths, chunk=3, overlap=1, processes=2))
    assert len(got) == len(want) == 6
    for a, b in zip(want, got):
        assert b.dtype == np.uint8 and np.array_equal(a, b)


def test_packed_sequence_feeds_the_same_chunks(tmp_path):
    """pack_sequence + FrameLoader(memory-mapped array): the decoded-once format yields the chunks the file list yields."""
    paths = loader.load_image_paths(os.path.join(GOLDEN, "kitti02")) * 2
    packed = loader.pack_sequence(paths, str(tmp_path / "seq.npy"))
    arr = np.load(packed, mmap_mode="r")
    assert arr.shape == (8, 376, 1241) and arr.dtype == np.uint8
    want = list(loader.FrameLoader(paths, chunk=3, overlap=1, workers=2))
    got = list(loader.FrameLoader(arr, chunk=3, overlap=1))
    assert len(got) == len(want)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        loader.FrameLoader(np.zeros((2, 3)), chunk=2)
