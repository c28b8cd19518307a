"""Pins include/mxdet_math.h (the only code shared by oracle and product) against independent references."""
import numpy as np


def test_philox_known_answer_vectors(oracle):
    # Random123 kat_vectors, philox4x32-10 (Salmon et al., SC'11)
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for c, k, want in kat:
        got = oracle.philox(c, k)
        assert tuple(int(v) for v in got) == want


def test_expf_logf_against_float64(oracle):
    xs = np.concatenate([np.linspace(-20, 20, 2001), np.array([0.0, 1e-6, -1e-6, 4.135166556742356])]).astype(np.float32)
    got = oracle.expf(xs).astype(np.float64)
    want = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - want) / want) < 4e-7
    ys = np.concatenate([np.exp(np.linspace(-12, 12, 2001)), np.array([1.0, 0.5, 2.0])]).astype(np.float32)
    got = oracle.logf(ys).astype(np.float64)
    want = np.log(ys.astype(np.float64))
    assert np.max(np.abs(got - want)) < 3e-7 * np.maximum(1.0, np.abs(want)).max()
    assert oracle.logf(np.float32(1.0)) == 0.0
    assert oracle.expf(np.float32(0.0)) == 1.0


def test_bf16_round_to_nearest_even(oracle):
    import torch
    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32) * 37.0
    want = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    got = oracle.f32_to_bf16_bits(x)
    assert np.array_equal(got, want)
    L = oracle.lib()
    for v in (1.0, 1.00390625, 1.01171875, -3.3895313892515355e38, 65504.0):
        assert L.oracle_f32_to_bf16(v) == int(torch.tensor([v]).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)[0])


def test_decode_encode_roundtrip_and_float64(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        ex = np.sort(rng.uniform(0, 500, 2)).tolist() + np.sort(rng.uniform(0, 500, 2)).tolist()
        ex = np.array([ex[0], ex[2], ex[1] + 8, ex[3] + 8], np.float32)
        gt = np.array([ex[0] + rng.uniform(-5, 5), ex[1] + rng.uniform(-5, 5), ex[2] + rng.uniform(-5, 5) + 3,
                       ex[3] + rng.uniform(-5, 5) + 3], np.float32)
        d = oracle.encode(ex, gt)
        back = oracle.decode_clip(ex, d, 10000.0, 10000.0)
        assert np.allclose(back, np.clip(gt, 0, 9999), atol=2e-3)
        # float64 restatement of the published transform
        ew, eh = ex[2] - ex[0] + 1.0, ex[3] - ex[1] + 1.0
        gw, gh = gt[2] - gt[0] + 1.0, gt[3] - gt[1] + 1.0
        want = [((gt[0] + 0.5 * (gw - 1)) - (ex[0] + 0.5 * (ew - 1))) / ew,
                ((gt[1] + 0.5 * (gh - 1)) - (ex[1] + 0.5 * (eh - 1))) / eh, np.log(gw / ew), np.log(gh / eh)]
        assert np.allclose(d, want, atol=1e-5)
