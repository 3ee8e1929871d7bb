"""Disk-format ingest: MV .bin reader and the ToTensor + Normalize restatement (CPU)."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref


def test_read_mv_bin_roundtrip(tmp_path):
    from arseg_amd import ingest

    g = np.random.Generator(np.random.PCG64(3))
    mv = (g.integers(-600, 601, (36, 48, 2)) * 4).astype(np.int16)          # integer-pel motion in quarter-pel units
    p = tmp_path / "0001.bin"
    mv.tofile(p)
    got = ingest.read_mv_bin(p, 36, 48)
    assert got.dtype == np.int16 and np.array_equal(got, mv)
    # the reference's read (dataset/camvid.py:624-626): np.fromfile(f, np.short).reshape(H, W, 2) / 4 -> float64 pixels
    assert np.array_equal(got.astype(np.float64) / 4, np.fromfile(p, np.dtype(np.short)).reshape(36, 48, 2) / 4)
    assert np.array_equal(cpu_ref.mv_from_int16(torch.from_numpy(got[None])).numpy()[0], got.astype(np.float64) / 4)
    with pytest.raises(ValueError):
        ingest.read_mv_bin(p, 36, 47)


def test_to_tensor_normalize_restatement():
    from arseg_amd import ingest

    g = np.random.Generator(np.random.PCG64(4))
    img = g.integers(0, 256, (2, 5, 7, 3), dtype=np.uint8)
    got = cpu_ref.to_tensor_normalize(img, ingest.CAMVID_MEAN, ingest.CAMVID_STD)
    want = (img.astype(np.float32).transpose(0, 3, 1, 2) / np.float32(255) - np.array(ingest.CAMVID_MEAN, np.float32).reshape(1, 3, 1, 1)) \
        / np.array(ingest.CAMVID_STD, np.float32).reshape(1, 3, 1, 1)
    assert got.shape == (2, 3, 5, 7) and got.dtype == torch.float32
    assert np.abs(got.numpy() - want).max() <= 1e-6
