"""GPU parity of the loader's image preparation against the oracle's
restatement of minibatch_wsl.py:121-157 + blob.py:67-131 (cv2.resize INTER_LINEAR): same
operations in the same order, correctly rounded -> bit-identical float32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
MEANS = (102.9801, 115.9465, 122.7717)


@pytest.mark.parametrize('scale', [1.0, 2.0, 0.5, 600.0 / 37.0, 0.731, 1.0 / 3.0])
@pytest.mark.parametrize('flip', [False, True])
def test_prep_image_matches_oracle(dev, scale, flip):
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(51)
    im = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for crop in (None, (3, 5, 30, 47), (0, 0, 0, 52), (36, 52, 36, 52)):
        ref = oracle.prep_image(im, scale, flip=flip, crop=crop, means=MEANS, stds=(1.0, 2.0, 0.5))
        oh, ow = ref.shape[:2]
        if oh == 0 or ow == 0:
            continue
        out = torch.full((3, oh + 2, ow + 3), 7.0, device=dev)
        got = ops.prep_image(torch.from_numpy(im).to(dev), out, scale, flip=flip, crop=crop,
                             means=MEANS, stds=(1.0, 2.0, 0.5))
        assert got == (oh, ow)
        o = out.cpu().numpy()
        assert np.array_equal(o[:, :oh, :ow], ref.transpose(2, 0, 1))
        assert (o[:, oh:, :] == 7.0).all() and (o[:, :, ow:] == 7.0).all()   # padding untouched


@pytest.mark.parametrize('distort', [(1.0, 1.0), (1.37, 0.71), (1 / 1.5, 1.5), (1.5, 1 / 1.5)])
def test_prep_image_hsv_distortion(dev, distort):
    """WSL.USE_DISTORTION on the GPU: the uint8 HSV round trip is integer / rounded-float work,
    held bit-exact against the oracle's restatement of cv2's 8-bit conversions."""
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(52)
    im = rng.integers(0, 256, (41, 67, 3), dtype=np.uint8)
    im[0, :7] = [[0, 0, 255], [0, 255, 0], [255, 0, 0], [255, 255, 255], [0, 0, 0], [7, 7, 7],
                 [255, 254, 255]]
    # scale 1, no mean: the output IS the distorted uint8 image
    ref = oracle.distort_hsv(im, *distort)
    out = torch.zeros((3, 41, 67), device=dev)
    ops.prep_image(torch.from_numpy(im).to(dev), out, 1.0, distort=distort)
    assert np.array_equal(out.cpu().numpy().transpose(1, 2, 0), ref.astype(np.float32))
    # and through flip + crop + resize
    ref = oracle.prep_image(im, 1.7, flip=True, crop=(2, 3, 38, 60), means=MEANS, distort=distort)
    out = torch.zeros((3,) + ref.shape[:2], device=dev)
    ops.prep_image(torch.from_numpy(im).to(dev), out, 1.7, flip=True, crop=(2, 3, 38, 60),
                   means=MEANS, distort=distort)
    assert np.array_equal(out.cpu().numpy(), ref.transpose(2, 0, 1))


def test_prep_image_errors(dev):
    from naws_hip import ops, lib
    im = torch.zeros((8, 8, 3), dtype=torch.uint8, device=dev)
    with pytest.raises(TypeError):
        ops.prep_image(im, torch.zeros((3, 4, 4), device=dev), 1.0)          # slot too small
    with pytest.raises(lib.NawsError):
        ops.prep_image(im, torch.zeros((3, 32, 32), device=dev), 1.0, crop=(0, 0, 8, 8))


def test_loader_device_prep_equals_host_path(dev, cfgmod):
    """NAWS.DEVICE_PREP: the batch the loader stages on the GPU equals the host-prepared batch
    (same RNG), including zero padding of B = 2 images of different sizes and a mixup pair."""
    import os
    c = cfgmod
    c.merge_cfg_from_file(os.path.join(os.path.dirname(__file__), '..', 'na-fwebsod_amd', 'configs',
                                       'flickr_voc', 'na_wsddn_V-16-C5_1x.yaml'))
    c.merge_cfg_from_list(['TRAIN.SCALES', '(48, 64)',
                           'TRAIN.MAX_SIZE', 100, 'NAWS.DEVICE_PREP', True])
    from detectron.datasets import synthetic
    from detectron.roi_data import minibatch_wsl, loader_wsl
    roidb = synthetic.make_roidb(3, 10, 20, 40, 64, seed=5)
    roidb[1]['flipped'] = True
    roidb[2]['height'], roidb[2]['width'] = 50, 44
    parts_h, parts_r = [], []
    for e in roidb[:2]:
        np.random.seed(9)
        parts_h.append(minibatch_wsl.get_minibatch([e], raw=False)[0])
        np.random.seed(9)
        parts_r.append(minibatch_wsl.get_minibatch([e])[0])
    np.random.seed(4)
    mh = loader_wsl.mixup_blobs(minibatch_wsl.get_minibatch(roidb[1:], raw=False)[0], 0.3)
    np.random.seed(4)
    mr = loader_wsl.mixup_blobs(minibatch_wsl.get_minibatch(roidb[1:])[0], 0.3)
    host = loader_wsl.collate(parts_h + [mh])
    raw = loader_wsl.collate(parts_r + [mr])
    data = loader_wsl.device_prep_images(raw['_raw'], raw['_mix'], dev)
    torch.cuda.synchronize()
    assert data.shape == host['data'].shape
    assert np.array_equal(data.cpu().numpy(), host['data'])


def test_pinned_ring_round_trip(dev):
    """The loader's staging ring: mixed dtypes / odd sizes through one copy, slots reused and
    regrown, views typed and shaped like their sources."""
    import sys
    from detectron.roi_data.loader_wsl import PinnedRing
    ring = PinnedRing(dev, slots=2)
    rng = np.random.default_rng(5)
    for it in range(7):
        n = 3 + it * 1000
        arrays = {'rois': rng.uniform(0, 500, (n, 5)).astype(np.float32),
                  'labels_int32': rng.integers(0, 20, (1, 1)).astype(np.int32),
                  ('_im', 0, 0): rng.integers(0, 256, (37 + it, 53, 3)).astype(np.uint8),
                  'ids': rng.integers(0, 1 << 40, (2,)).astype(np.int64),
                  'strided': rng.standard_normal((8, 6)).astype(np.float32)[:, ::2]}
        if it == 5:
            arrays['big'] = rng.standard_normal((600, 1000)).astype(np.float32)   # grows the slot
        out = ring.upload(arrays)
        for k, v in arrays.items():
            t = out[k]
            assert tuple(t.shape) == v.shape and str(t.dtype).split('.')[-1] == v.dtype.name
            assert np.array_equal(t.cpu().numpy(), v), k
