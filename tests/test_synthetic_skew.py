"""synthetic.skew_blobs / skew_images (the non-Kaiming statistics of the full-size parity tests
and of bench.py's in-run check): the skewed network is the SAME function as the un-skewed one
with the same biases - checked through the oracle on a small image - while its weights and
activations have the advertised spread."""
import numpy as np


def test_skewed_network_is_the_same_function():
    from detectron.datasets import synthetic
    from oracle import oracle
    c = 20
    blobs = synthetic.init_blobs(c, seed=3)
    mb = synthetic.make_minibatch(synthetic.make_roidb(1, 32, c, 96, 128, seed=3), c)
    plain = mb['data'].copy()
    mb['data'] = synthetic.skew_images(mb['data'])
    third = plain.shape[3] // 3
    assert np.abs(mb['data'][..., :third]).max() <= (np.abs(plain).max() + 40.0) * 2.0 ** -12
    assert np.abs(mb['data'][..., third:]).max() > 100.0
    flat = synthetic.skew_blobs(blobs, seed=3, span=0.0)       # same biases, no channel factors
    skew = synthetic.skew_blobs(blobs, seed=3, span=6.0)
    assert float(flat['conv3_2_b'].abs().max()) > 0 and float(flat['fc7_b'].abs().max()) > 0
    for k in ('conv1_2_w', 'conv4_3_w', 'fc6_w', '_[noisy]_fc7_w'):
        w = skew[k].reshape(skew[k].shape[0], -1)
        rows = w.abs().max(dim=1).values
        assert float(rows.max() / rows.min()) > 2.0 ** 8, k      # output channels: up to 2^12
        assert float((rows / w.abs().median(dim=1).values).max()) > 2.0 ** 5, k   # inside a row
    r0 = oracle.full_forward_backward(flat, mb, None, c, train=False)
    r1 = oracle.full_forward_backward(skew, mb, None, c, train=False)
    for k in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d'):
        a, b = r0['act'][k], r1['act'][k]
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max(), k
    cm = np.abs(r1['conv5_3']).max(axis=(0, 2, 3))
    cm = cm[cm > 0]
    assert cm.max() / cm.min() > 2.0 ** 8                        # activations: per-channel spread
    np.testing.assert_allclose(r1['tails'][0]['loss_cls'], r0['tails'][0]['loss_cls'], rtol=1e-4)
