"""
GPU tests of the long block's MDCT (mdct_long_kernel, mdct.py:63-76 / window.py:104-121 / codecThem.py:363-364) over
every layout the encoder hands it -- hop-overlapped streams, explicit offsets a hop apart / in runs / unordered / at odd
sample offsets, blocks at stride 2048, channel bases on odd samples; mono and joint; int16 PCM and float64 samples;
counts that are no multiple of a wave's run.  The lines of every layout are checked against the oracle's
(|dX| <= 1e-12 max|X|), and the layouts against each other BIT FOR BIT: the kernel folds and transforms a block the same
way wherever its samples came from, so the same block must give the same float64 lines in every layout.
Everything goes through the C ABI (mrc_dev_encode_ex with a lines buffer).
"""
import numpy as np
import pytest

from oracle import fast

pytestmark = pytest.mark.gpu
MDCT_RTOL = 1e-12
HOP = 1024


@pytest.fixture(scope="module")
def h():
    from mrcaudiocodec_amd import Handle
    hd = Handle(device_id=0)
    yield hd
    hd.close()


def _streams(torch, hops, seed=11):
    rng = np.random.default_rng(seed)
    left = np.clip(np.rint(rng.normal(0, 5000, (hops + 2) * HOP)), -32768, 32767).astype(np.int16)
    right = np.clip(np.rint(0.7 * left + rng.normal(0, 900, left.shape)), -32768, 32767).astype(np.int16)
    left[HOP:HOP + 16] = -32768                      # the code without a positive twin (pcmfile.py:91-100 maps it to -0)
    return left, right


def _lines(torch, enc, left, right, n, stride, offsets):
    nsig = 4 if right is not None else 1
    lines = torch.full((n * nsig * HOP,), float("nan"), dtype=torch.float64, device=left.device)
    out = enc.encode(1024, 1024, left, right, n, stride, offsets, lines_out=lines, fresh=True)
    torch.cuda.synchronize()
    return lines.cpu().numpy().reshape(n, nsig, HOP), {k: v.cpu().numpy() for k, v in out.items()}


def _oracle_lines(fl, fr, starts):
    """lines of the blocks at the given sample offsets: [n, nsig, 1024] (L, R, M, S as codecThem.py:363-364 forms them)"""
    bl = np.stack([fl[s:s + 2048] for s in starts])
    if fr is None:
        return fast.mdct_batch(bl, 1024, 1024)[:, None, :]
    br = np.stack([fr[s:s + 2048] for s in starts])
    sig = (bl, br, (bl + br) / 2.0, (bl - br) / 2.0)
    return np.stack([fast.mdct_batch(x, 1024, 1024) for x in sig], axis=1)


@pytest.mark.parametrize("fmt", ["i16", "f64"])
def test_long_mdct_layouts_agree_bitwise_and_match_the_oracle(h, fmt):
    torch = pytest.importorskip("torch")
    from mrcaudiocodec_amd import synth
    from mrcaudiocodec_amd.batch import StreamEncoder
    enc = StreamEncoder(handle=h)
    hops = 150
    l16, r16 = _streams(torch, hops)
    fl, fr = synth.pcm_to_float(l16), synth.pcm_to_float(r16)
    if fmt == "i16":
        dl, dr = torch.from_numpy(l16).to("cuda:0"), torch.from_numpy(r16).to("cuda:0")
    else:
        dl, dr = torch.from_numpy(fl).to("cuda:0"), torch.from_numpy(fr).to("cuda:0")

    def offs(v):
        return torch.tensor(np.asarray(v, dtype=np.int64), device="cuda:0")

    n = 37                                                        # no multiple of 16 (a wave's run) nor of 8 (a joint half run)
    for joint in (False, True):
        R, Rf = (dr, fr) if joint else (None, None)
        # hop-overlapped stream == explicit offsets a hop apart == the oracle
        starts = np.arange(n) * HOP
        X_stream, i_stream = _lines(torch, enc, dl, R, n, HOP, None)
        X_offs, i_offs = _lines(torch, enc, dl, R, n, 0, offs(starts))
        ref = _oracle_lines(fl, Rf, starts)
        assert np.abs(X_stream - ref).max() <= MDCT_RTOL * np.abs(ref).max()
        assert np.array_equal(X_stream, X_offs)
        for k in i_stream:
            assert np.array_equal(i_stream[k], i_offs[k]), k
        # runs of four with a hop skipped, unordered blocks of mixed parity, odd sample offsets: each against the oracle and,
        # block by block, against the same block of the stream layout where there is one
        for starts in (((np.arange(n) // 4) * 5 + np.arange(n) % 4) * HOP,
                       np.random.default_rng(2).permutation(60)[:n] * 2048 + np.arange(n) % 7,
                       np.arange(n) * HOP + 333):
            X, _ = _lines(torch, enc, dl, R, n, 0, offs(starts))
            ref = _oracle_lines(fl, Rf, starts)
            assert np.abs(X - ref).max() <= MDCT_RTOL * np.abs(ref).max()
            for i, s in enumerate(starts):
                if s % HOP == 0 and s // HOP < n:
                    assert np.array_equal(X[i], X_stream[s // HOP]), (joint, int(s))
        # blocks at stride 2048 (the four waves of a workgroup on adjacent units, nothing kept between blocks)
        m = 29
        X, _ = _lines(torch, enc, dl, R, m, 2048, None)
        ref = _oracle_lines(fl, Rf, np.arange(m) * 2048)
        assert np.abs(X - ref).max() <= MDCT_RTOL * np.abs(ref).max()
        for i in range(m):
            if 2 * i < n:
                assert np.array_equal(X[i], X_stream[2 * i]), (joint, i)
        # channel bases on odd samples (int16: the base in the upper half of its 32-bit word) == the same blocks by odd offsets
        X_base, _ = _lines(torch, enc, dl[1:], None if R is None else R[1:], n, HOP, None)
        X_odd, _ = _lines(torch, enc, dl, R, n, 0, offs(np.arange(n) * HOP + 1))
        assert np.array_equal(X_base, X_odd)
        if joint:
            # bases of different parity: L on an odd sample, R on an even one -- against the oracle
            X_mix, _ = _lines(torch, enc, dl[1:], R[2:], n, HOP, None)
            bl = np.stack([fl[1 + i * HOP:1 + i * HOP + 2048] for i in range(n)])
            br = np.stack([fr[2 + i * HOP:2 + i * HOP + 2048] for i in range(n)])
            ref = np.stack([fast.mdct_batch(x, 1024, 1024) for x in (bl, br, (bl + br) / 2.0, (bl - br) / 2.0)], axis=1)
            assert np.abs(X_mix - ref).max() <= MDCT_RTOL * np.abs(ref).max()


def test_joint_lines_of_l_and_r_equal_the_mono_lines(h):
    # L and R of a joint block take the single-channel path of the kernel, M and S the two-channel one: the first two must
    # equal the mono transform of each channel bit for bit, and S of identical channels is exactly zero
    torch = pytest.importorskip("torch")
    from mrcaudiocodec_amd.batch import StreamEncoder
    enc = StreamEncoder(handle=h)
    n = 70
    l16, r16 = _streams(torch, n, seed=12)
    dl, dr = torch.from_numpy(l16).to("cuda:0"), torch.from_numpy(r16).to("cuda:0")
    Xj, _ = _lines(torch, enc, dl, dr, n, HOP, None)
    Xl, _ = _lines(torch, enc, dl, None, n, HOP, None)
    Xr, _ = _lines(torch, enc, dr, None, n, HOP, None)
    assert np.array_equal(Xj[:, 0], Xl[:, 0]) and np.array_equal(Xj[:, 1], Xr[:, 0])
    Xs, _ = _lines(torch, enc, dl, dl.clone(), n, HOP, None)
    assert np.array_equal(Xs[:, 2], Xl[:, 0]) and not Xs[:, 3].any()


@pytest.mark.parametrize("ab", [(128, 128), (1024, 128), (128, 1024)])
def test_wave_mdct_run_geometry_at_the_round_boundaries(h, ab):
    # mdct_wave_kernel (short / transition blocks) deals its groups of units to a whole number of rounds of wavefronts in
    # balanced runs of at most 16 groups (3 workgroups x 4 waves per CU x 256 CUs = 3072 wavefronts per round): counts on
    # both sides of "one group per wave" (3072 groups) and of "one round of runs of 16" (49 152 groups), for the transition
    # blocks' register-fold form and the short blocks' four-units-per-wave form.  The whole batch in one launch must equal,
    # bit for bit, the same blocks launched in pieces of 1000 (another geometry: one group per wave), and its first and last
    # blocks the oracle's lines; the integers of the encode ride along.
    torch = pytest.importorskip("torch")
    from mrcaudiocodec_amd import synth
    from mrcaudiocodec_amd.batch import StreamEncoder
    enc = StreamEncoder(handle=h)
    a, b = ab
    N, M = a + b, (a + b) // 2
    per_group = 4 if N == 256 else 1
    hops = 700
    l16, r16 = _streams(torch, hops, seed=23)
    fl = synth.pcm_to_float(l16)
    dl, dr = torch.from_numpy(l16).to("cuda:0"), torch.from_numpy(r16).to("cuda:0")
    span = hops * HOP - N

    def run(n, lo, hi, right=None):
        step = max(1, span // n)
        o = torch.arange(lo, hi, device="cuda:0", dtype=torch.int64) * step + 3        # (odd sample offsets too)
        nsig = 4 if right is not None else 1
        lines = torch.full(((hi - lo) * nsig * M,), float("nan"), dtype=torch.float64, device="cuda:0")
        out = enc.encode(a, b, dl, right, hi - lo, 0, o.contiguous(), lines_out=lines, fresh=True)
        torch.cuda.synchronize()
        return lines.cpu().numpy().reshape(hi - lo, nsig, M), {k: v.cpu().numpy() for k, v in out.items()}, step

    for groups in (3071, 3072, 3073, 49152, 49153):
        n = groups * per_group - (1 if groups % 2 else 0) * (per_group - 1)            # (odd counts end inside a group)
        X, ints, step = run(n, 0, n)
        assert not np.isnan(X).any()
        ends = list(range(4)) + list(range(n - 4, n))
        bl = np.stack([fl[i * step + 3:i * step + 3 + N] for i in ends])
        ref = fast.mdct_batch(bl, a, b)
        assert np.abs(X[ends, 0] - ref).max() <= MDCT_RTOL * np.abs(ref).max()
        for lo in range(0, n, 1000 * per_group):
            hi = min(n, lo + 1000 * per_group)
            # the piece uses the whole batch's step, so that its blocks are the same samples
            o = torch.arange(lo, hi, device="cuda:0", dtype=torch.int64) * step + 3
            lines = torch.full(((hi - lo) * M,), float("nan"), dtype=torch.float64, device="cuda:0")
            part = enc.encode(a, b, dl, None, hi - lo, 0, o.contiguous(), lines_out=lines, fresh=True)
            torch.cuda.synchronize()
            assert np.array_equal(lines.cpu().numpy().reshape(hi - lo, M), X[lo:hi, 0]), (groups, lo)
            for k in part:
                assert np.array_equal(part[k].cpu().numpy(), ints[k][lo:hi]), (groups, lo, k)
    # joint: units = 4 x blocks (L, R, M, S of a block are consecutive units of consecutive groups)
    fr = synth.pcm_to_float(r16)
    for nj in (769, 12289):
        Xj, _, step = run(nj, 0, nj, right=dr)
        assert not np.isnan(Xj).any()
        for i in (0, 1, nj // 2, nj - 1):
            bl, br = fl[i * step + 3:i * step + 3 + N][None], fr[i * step + 3:i * step + 3 + N][None]
            for q, x in enumerate((bl, br, (bl + br) / 2.0, (bl - br) / 2.0)):
                ref = fast.mdct_batch(x, a, b)
                assert np.abs(Xj[i, q] - ref[0]).max() <= MDCT_RTOL * max(np.abs(ref).max(), 1e-300), (nj, i, q)
