"""SURVEY 8(f) N1 -- FeCo on the native engine: k-means ids bit-exact against the contract restatement
(oracle/feco.py), cluster means and their gradient against the reference's step under torch autograd, and the
hand-chained gradient wav -> MFCC -> FeCo -> CMVN -> TDNN -> loss against the oracle's autograd."""
import numpy as np
import pytest
import torch

from conftest import log

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def hip_model(xv_weights):
    from speakerguard_amd.model.xv_plda import xv_plda
    return xv_plda.from_weights(xv_weights, device=DEV, dither=0.0)


@pytest.fixture(scope="module")
def oracle_model(xv_weights):
    from oracle.xv_plda import XvPlda
    return XvPlda(xv_weights, threshold=None)


def _standalone_compress(feat, ids, k):
    """sg_feco_compress (feature_level.py:204-216 given the ids) as its own launch."""
    from speakerguard_amd import _native as N
    from speakerguard_amd.metric.metric import _context
    B, F, D = feat.shape
    out = torch.empty(B, k, D, device=DEV)
    counts = torch.empty(B, k, device=DEV, dtype=torch.int32)
    _context(DEV).call("sg_feco_compress", N._ptr(feat), N._ptr(ids), B, F, D, k, N._ptr(out), N._ptr(counts),
                       N.current_stream_ptr(DEV))
    return out, counts


def _ids(feat, ratio=0.5, max_iter=10):
    from speakerguard_amd.defense.feature_level import FeCoDefense
    feat = feat.to(DEV).contiguous()
    out, (ids, counts, dims, force, keep) = FeCoDefense(ratio, max_iter=max_iter).fwd(feat)
    if keep is None:  # the means the clustering launch hands out are the standalone compress step of its ids, bit for bit
        out2, counts2 = _standalone_compress(feat, ids, dims[3])
        assert torch.equal(out, out2) and torch.equal(counts, counts2)
    return out, ids.cpu().numpy(), counts.cpu().numpy()


def test_kmeans_ids_bit_exact(hip_model):
    from oracle import feco
    from speakerguard_amd import synth
    rs = np.random.RandomState(4)
    mfcc = hip_model.compute_feat(torch.from_numpy(synth.make_waveforms(3, 48000, seed=70)).to(DEV), flag=1).cpu()
    cases = [("mfcc 3x300x30", mfcc, 0.5), ("random 2x77x13", torch.from_numpy(rs.randn(2, 77, 13).astype(np.float32)), 0.3),
             ("logmel-like 2x300x32", torch.from_numpy((rs.randn(2, 300, 32) * 10 - 40).astype(np.float32)), 0.5),
             ("duplicates", torch.from_numpy(np.repeat(rs.randn(1, 20, 8).astype(np.float32), 4, axis=1)), 0.5),
             # 15 s utterance: frames + centroids exceed one block's LDS -> the frames-in-HBM form of the kernel
             ("long 2x1500x30", torch.from_numpy((rs.randn(2, 1500, 30) * 3).astype(np.float32)), 0.5),
             # more than 32 dimensions (64-column operands, one centroid row per wave in the update), close to the
             # longest utterance one block accepts (slow member lists, no merge pass), a mid-size one (frames in LDS, slow lists
             # or not depending on what fits), few clusters, and an offset far from zero (the centring matters)
             ("wide 2x200x48", torch.from_numpy((rs.randn(2, 200, 48) * 2).astype(np.float32)), 0.5),
             ("wide 1x90x64", torch.from_numpy(rs.randn(1, 90, 64).astype(np.float32)), 0.4),
             ("longest 1x1850x32", torch.from_numpy((rs.randn(1, 1850, 32) * 5 - 30).astype(np.float32)), 0.5),
             ("mid 2x700x32", torch.from_numpy((rs.randn(2, 700, 32) * 5).astype(np.float32)), 0.5),
             ("mid 1x1000x20", torch.from_numpy((rs.randn(1, 1000, 20) * 5).astype(np.float32)), 0.25),
             ("offset 2x300x32", torch.from_numpy((rs.randn(2, 300, 32) * 4 + 300).astype(np.float32)), 0.5),
             ("few clusters 3x300x32", torch.from_numpy((rs.randn(3, 300, 32) * 4).astype(np.float32)), 0.04)]
    for name, feat, ratio in cases:
        _, ids, counts = _ids(feat, ratio)
        k = int(feat.shape[1] * ratio)
        for b in range(feat.shape[0]):
            want = feco.kmeans_ids(feat[b].numpy(), k)
            assert np.array_equal(ids[b], want), "%s utt %d: %d ids differ" % (name, b, (ids[b] != want).sum())
            assert np.array_equal(counts[b], np.bincount(want, minlength=k))
        log("FeCo k-means %s: ids bit-exact, empty clusters %d" % (name, int((counts == 0).sum())))


def test_seeded_random_init_matches_restatement_and_is_keyed_by_position():
    """The randomised form of the defense (what EOT attacks average over; the reference's k-means starts from a random
    draw of numpy's global generator): initial frames from Philox4x32-10 keyed by (seed, call, GLOBAL utterance) --
    ids bit-exact against oracle.feco.kmeans_ids started from oracle.philox.feco_random_init; a second call draws
    afresh; a shard that names its offset reproduces its rows of the full batch."""
    from oracle import feco, philox
    from speakerguard_amd.defense.feature_level import FeCoDefense
    rs = np.random.RandomState(9)
    for name, feat, ratio in [("logmel-like 3x300x32", torch.from_numpy((rs.randn(3, 300, 32) * 10 - 40).astype(np.float32)), 0.5),
                              ("random 2x77x13", torch.from_numpy(rs.randn(2, 77, 13).astype(np.float32)), 0.3),
                              ("long 1x1500x30", torch.from_numpy((rs.randn(1, 1500, 30) * 3).astype(np.float32)), 0.5)]:
        B, F, _ = feat.shape
        k = int(F * ratio)
        d = FeCoDefense(ratio, init='random', seed=21)
        d.index_base = 5
        per_call = []
        for call in range(2):
            out, (ids, counts, _, _, _) = d.fwd(feat.to(DEV))
            out2, counts2 = _standalone_compress(feat.to(DEV).contiguous(), ids, k)
            assert torch.equal(out, out2) and torch.equal(counts, counts2)
            ids = ids.cpu().numpy()
            for b in range(B):
                init = philox.feco_random_init(d.call_seed(call), 5 + b, F, k)
                assert len(set(init.tolist())) == k
                want = feco.kmeans_ids(feat[b].numpy(), k, init_frames=init)
                assert np.array_equal(ids[b], want), "%s call %d utt %d: %d ids differ" % (name, call, b, (ids[b] != want).sum())
            per_call.append(ids)
        assert not np.array_equal(per_call[0], per_call[1]), "every call of the randomised defense draws fresh frames"
        if B > 1:  # rows 1.. attacked as their own shard
            tail = FeCoDefense(ratio, init='random', seed=21)
            tail.index_base = 6
            _, (ids_t, _, _, _, _) = tail.fwd(feat[1:].to(DEV))
            assert np.array_equal(ids_t.cpu().numpy(), per_call[0][1:])
        even = _ids(feat, ratio)[1]
        assert not np.array_equal(even, per_call[0])
        log("FeCo k-means, seeded random init %s: ids bit-exact vs Philox restatement; fresh per call; shard-invariant" % name)


def test_two_compute_units_per_instance_give_the_same_clustering():
    """Round 5: an instance's k-means runs on two compute units when the batch leaves room (both blocks compute the same
    clustering and share the assignment step's work, sg_feco_set_two_cu).  Same ids, means and counts as one block per
    instance -- and as a run whose second blocks publish NOTHING (fault injection): the first blocks time out after 20 us and
    compute everything themselves, the second ones find the give-up mark and do the same."""
    from speakerguard_amd import _native as N
    from speakerguard_amd.metric.metric import _context
    ctx = _context(DEV)
    rs = np.random.RandomState(31)
    for B, F, D, reps in ((64, 300, 32, 2), (5, 300, 30, 1), (3, 130, 13, 4)):
        feat = torch.from_numpy((rs.randn(B, F, D) * 10 - 40).astype(np.float32)).to(DEV)
        k = F // 2
        got = {}
        for mode in ("one", "two", "two, partner silent"):
            ctx.call("sg_feco_set_two_cu", 0 if mode == "one" else -1)
            if mode.endswith("silent"):
                ctx.call("sg_debug_lose_handoffs", 1)
            ids = torch.empty(reps * B, F, device=DEV, dtype=torch.int32)
            out = torch.empty(reps * B, k, D, device=DEV)
            counts = torch.empty(reps * B, k, device=DEV, dtype=torch.int32)
            ctx.call("sg_feco_kmeans_compress", N._ptr(feat), B, F, D, k, 10, 1, 77, 3, reps, N._ptr(ids), N._ptr(out), N._ptr(counts),
                     N.current_stream_ptr(DEV))
            torch.cuda.synchronize()
            got[mode] = (ids.clone(), out.clone(), counts.clone())
        ctx.call("sg_debug_lose_handoffs", 0)
        ctx.call("sg_feco_set_two_cu", -1)
        for mode in ("two", "two, partner silent"):
            for a, b in zip(got["one"], got[mode]):
                assert torch.equal(a, b), (B, F, D, reps, mode)
        assert got["one"][0].min().item() >= 0 and got["one"][2].sum().item() == reps * B * F
    # the exchange words carry 15 bits of launch counter: past 2^15 launches of one context the buffers are wiped and the
    # tags start over.  The counter is parked just below the wrap (test hook sg_debug_feco_epoch; until round 5 this test
    # spent 33 000 real launches to get there): the launches above used the tags 1, 2, ..., the launches after the wrap use
    # them again, with the words of the earlier cycle still in the buffer if the wipe did not happen -- and the answer is
    # still the one-block one.  Same across the wrap of the 32-bit counter itself (0 is never a launch id).
    feat = torch.from_numpy(rs.randn(2, 96, 8).astype(np.float32)).to(DEV)
    ids = torch.empty(2, 96, device=DEV, dtype=torch.int32)
    out = torch.empty(2, 48, 8, device=DEV)
    counts = torch.empty(2, 48, device=DEV, dtype=torch.int32)
    args = (N._ptr(feat), 2, 96, 8, 48, 10, 1, 5, 0, 1, N._ptr(ids), N._ptr(out), N._ptr(counts), N.current_stream_ptr(DEV))
    ctx.call("sg_feco_set_two_cu", 0)
    ctx.call("sg_feco_kmeans_compress", *args)
    torch.cuda.synchronize()
    want = (ids.clone(), out.clone(), counts.clone())
    ctx.call("sg_feco_set_two_cu", -1)
    for i in range(24):  # a first cycle's words with the tags 1 .. 24 (+ whatever the shapes above left)
        ctx.call("sg_feco_kmeans_compress", *args)
    for park in (0x7FF4, 0xFFF4, 0xFFFFFFF4):
        ctx.call("sg_debug_feco_epoch", park)
        for i in range(40):
            ids.fill_(-1)
            ctx.call("sg_feco_kmeans_compress", *args)
            torch.cuda.synchronize()
            assert torch.equal(ids, want[0]) and torch.equal(out, want[1]) and torch.equal(counts, want[2]), (hex(park), i)
    log("FeCo k-means on two compute units per instance: ids, means, counts equal to one block per instance, also with a silent partner "
        "and across the wraps of the exchange tags and of the launch counter")


def test_kmeans_refuses_what_does_not_fit():
    from speakerguard_amd import _native as N
    from speakerguard_amd.defense.feature_level import FeCoDefense
    with pytest.raises(N.NativeError, match="too long"):
        FeCoDefense(0.5).fwd(torch.zeros(1, 4000, 32, device=DEV))   # 2000 centroids x 32 dims = 256 KB > one block's LDS


def test_compress_forward_backward_match_reference_step():
    """Given ids (incl. empty clusters): forward = per-cluster torch.mean / fallback, backward = its autograd."""
    from oracle import feco
    from speakerguard_amd import _native as N
    from speakerguard_amd.metric.metric import _context
    rs = np.random.RandomState(5)
    B, F, D, k = 3, 40, 6, 12
    feat = torch.from_numpy(rs.randn(B, F, D).astype(np.float32))
    ids = rs.randint(0, k, size=(B, F)).astype(np.int32)
    ids[ids == 3] = 4  # cluster 3 empty everywhere
    ids[1][ids[1] == 7] = 8
    g = torch.from_numpy(rs.randn(B, k, D).astype(np.float32))
    fd, idd, gd = feat.to(DEV), torch.from_numpy(ids).to(DEV), g.to(DEV)
    out = torch.empty(B, k, D, device=DEV)
    counts = torch.empty(B, k, device=DEV, dtype=torch.int32)
    dfe = torch.empty(B, F, D, device=DEV)
    ctx, s = _context(DEV), N.current_stream_ptr(DEV)
    ctx.call("sg_feco_compress", N._ptr(fd), N._ptr(idd), B, F, D, k, N._ptr(out), N._ptr(counts), s)
    ctx.call("sg_feco_compress_backward", N._ptr(gd), N._ptr(idd), N._ptr(counts), B, F, D, k, 1, N._ptr(dfe), s)
    x = feat.clone().requires_grad_(True)
    want = torch.stack([feco.compress_from_ids(x[b], ids[b], k, force=True) for b in range(B)])
    (want * g).sum().backward()
    assert (out.cpu() - want.detach()).abs().max().item() < 1e-6
    assert (dfe.cpu() - x.grad).abs().max().item() < 1e-6
    assert int(counts[0, 3]) == 0 and torch.equal(out[0, 3].cpu(), feat[0, 3])  # fallback row = frame i


def test_compress_given_ids_matches_the_reference_run():
    """The HIP step after the clustering against tests/golden/feco_ref.npz: the reference's OWN FEATURE_COMPRESSION / kmeans
    code (feature_level.py:21-50,168-217) on given ids -- per-cluster means, the empty-cluster fallback of a batch
    (force), the DROP of empty clusters for a single utterance, and autograd's gradient of each.  Tolerance: the device
    sums a cluster's members in ascending frame order, torch.mean reduces pairwise: <= 2 ulp of the largest member sum."""
    import os
    from speakerguard_amd.defense.feature_level import FeCoDefense
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "feco_ref.npz"))
    worst = {}
    for tag in ("mfcc_b3", "logmel_b2", "single_drop", "single_full"):
        feat = torch.from_numpy(z[tag + "_feat"]).to(DEV)
        ids = torch.from_numpy(z[tag + "_ids"]).to(DEV)
        d = FeCoDefense(float(z[tag + "_ratio"]))
        out, saved = d.fwd(feat, ids=ids)
        want = torch.from_numpy(z[tag + "_out"])
        assert tuple(out.shape) == tuple(want.shape), (tag, out.shape, want.shape)  # incl. (1, k - 3, D) for the drop case
        e_out = (out.cpu() - want).abs().max().item()
        got = d.bwd(saved, torch.from_numpy(z[tag + "_cot"]).to(DEV))
        e_grad = (got.cpu() - torch.from_numpy(z[tag + "_dfeat"])).abs().max().item()
        worst[tag] = (e_out, e_grad)
        assert e_out < 2e-6 and e_grad < 1e-6, (tag, e_out, e_grad)
    # a fallback row is a copy of frame i, a singleton cluster a copy of its member: exact
    out, _ = FeCoDefense(0.5).fwd(torch.from_numpy(z["mfcc_b3_feat"]).to(DEV), ids=torch.from_numpy(z["mfcc_b3_ids"]).to(DEV))
    assert torch.equal(out[1, 3].cpu(), torch.from_numpy(z["mfcc_b3_feat"][1, 3]))
    log("FeCo given ids vs the reference run (max abs err out, grad): %s" % worst)


def test_single_utterance_drops_empty_clusters():
    from oracle import feco
    from speakerguard_amd.defense.feature_level import FeCoDefense
    rs = np.random.RandomState(6)
    base = rs.randn(1, 10, 4).astype(np.float32)
    feat = torch.from_numpy(np.repeat(base, 3, axis=1))  # 30 frames, 10 distinct -> k = 15 has empty clusters
    d = FeCoDefense(0.5)
    out, saved = d.fwd(feat.to(DEV))
    ids = saved[0].cpu().numpy()[0]
    x = feat[0].clone().requires_grad_(True)
    want = feco.compress_from_ids(x, ids, 15, force=False)
    assert out.shape == (1,) + tuple(want.shape) and want.shape[0] < 15
    assert (out[0].cpu() - want.detach()).abs().max().item() < 1e-6
    g = torch.from_numpy(rs.randn(*want.shape).astype(np.float32))
    (want * g).sum().backward()
    got = d.bwd(saved, g.unsqueeze(0).to(DEV))
    assert (got[0].cpu() - x.grad).abs().max().item() < 1e-6


@pytest.mark.parametrize("level", [1, 2])
def test_gradient_through_feco_matches_oracle_autograd(hip_model, oracle_model, level):
    """d loss / d wav through MFCC -> [FeCo at flag 1 | CMVN -> FeCo at flag 2] -> ... vs the oracle's autograd.
    The oracle is given the ids the device computed (the clustering is piecewise constant; its own ids are also
    compared and reported)."""
    from oracle import attacks as oatk
    from oracle import feco
    from speakerguard_amd import synth
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    from speakerguard_amd.defense.feature_level import FeCoDefense
    from speakerguard_amd.model.defended_model import defended_model
    x = torch.from_numpy(synth.make_waveforms(3, 32000, seed=71))
    d = FeCoDefense(0.5)
    dm = defended_model(hip_model, defense=[(level, d)])
    with torch.no_grad():
        y = oracle_model.make_decision(x)[0]
    dec, scores, loss, grad = dm.loss_grad(x.to(DEV), y.to(DEV), SEC4SR_CrossEntropy())
    # the same forward through the public (forward-only) path
    dec2, scores2 = dm.make_decision(x.to(DEV))
    assert torch.equal(dec, dec2) and torch.equal(scores, scores2)
    feats_dev = hip_model.compute_feat(x.to(DEV), flag=level)
    ids_dev = d.fwd(feats_dev)[1][0].cpu().numpy()
    xin = x.clone().requires_grad_(True)
    feats = oracle_model.compute_feat(xin, flag=level)
    k = feats.shape[1] // 2
    same = np.mean([np.mean(feco.kmeans_ids(feats[b].detach().numpy(), k) == ids_dev[b]) for b in range(3)])
    comp = torch.stack([feco.compress_from_ids(feats[b], ids_dev[b], k, force=True) for b in range(3)])
    _, sc = oracle_model.make_decision(comp, flag=level)
    lo = oatk.cross_entropy_loss(sc, y)
    lo.backward(torch.ones(3))
    want, got = xin.grad.numpy(), grad.cpu().numpy()
    gs = np.abs(want).max()
    err = np.abs(got - want).max() / gs
    log("FeCo at flag %d: oracle-feature ids equal to device ids %.4f; wav grad err/max %.3e; score err %.3e" % (
        level, same, err, (scores.cpu() - sc.detach()).abs().max().item()))
    assert dec.cpu().tolist() == sc.argmax(1).tolist()
    assert (scores.cpu() - sc.detach()).abs().max().item() < 5e-3
    # Stage by stage (tests/tools/feco_debug.py) the FeCo and MFCC backward are exact given the oracle's upstream
    # gradient (0 and 2.5e-6); what remains is the TDNN's own fp32 behaviour: one ReLU whose pre-activation is
    # within round-off of 0 flips between the two implementations and changes the gradient of its receptive
    # field (seen at flag 2: 0.1 % of the samples, 6e-3 of max).  Hence a bulk tolerance plus a bound on outliers.
    bad = float((np.abs(got - want) > 3e-3 * gs).mean())
    assert bad < 5e-3 and err < 2e-2, (bad, err)
    assert same > 0.97


def test_bpda_input_defense_in_the_gradient_chain(hip_model):
    """BASELINE configs[3] names "EOT/BPDA": a non-differentiable input transform wrapped in the reference's
    straight-through BPDA (BPDA.py:7-65, used for QT at defense/time_domain.py:44) has gradient identity, so
    d loss/d x of the defended model at x is the base model's gradient AT the transformed input -- alone, and stacked
    in front of a feature-level FeCo."""
    from speakerguard_amd import synth
    from speakerguard_amd.adaptive_attack.BPDA import straight_through
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    from speakerguard_amd.defense.feature_level import FeCoDefense
    from speakerguard_amd.model.defended_model import defended_model
    x = torch.from_numpy(synth.make_waveforms(3, 24000, seed=74)).to(DEV)
    quant = straight_through(lambda a: torch.round(a * 32768.0 / 256.0) * 256.0 / 32768.0)   # 8-bit depth reduction
    spec = SEC4SR_CrossEntropy()
    y = hip_model.make_decision(x)[0]
    dm = defended_model(hip_model, defense=[(0, quant)])
    dec, scores, loss, grad = dm.loss_grad(x, y, spec)
    d0, s0, l0, g0 = hip_model.loss_grad(quant(x), y, spec)
    assert torch.equal(grad, g0) and torch.equal(scores, s0) and torch.equal(loss, l0)
    assert torch.equal(dm.make_decision(x)[1], s0)
    # stacked: quantisation (flag 0) then FeCo on the raw features (flag 1)
    feco = FeCoDefense(0.5)
    dm2 = defended_model(hip_model, defense=[(0, quant), (1, feco)])
    ref = defended_model(hip_model, defense=[(1, feco)])
    _, s2, _, g2 = dm2.loss_grad(x, y, spec)
    _, s2r, _, g2r = ref.loss_grad(quant(x), y, spec)
    assert torch.equal(s2, s2r) and torch.equal(g2, g2r)
    # and the attack loop runs against it (step path: a defense sits between attack and model)
    adv, success = PGD(dm2, epsilon=0.002, step_size=0.0004, max_iter=3, batch_size=3, EOT_size=2, EOT_batch_size=1,
                       verbose=0).attack(x, y)
    assert (adv - x).abs().max().item() <= 0.002 + 1e-7 and len(success) == 3
    log("BPDA(quantise) + FeCo in the gradient chain: identity backward verified, PGD+EOT ran, success %s" % success)


def test_pgd_against_feco_defended_model(hip_model):
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.defense.feature_level import FeCo, FeCoDefense
    from speakerguard_amd.model.defended_model import defended_model
    x = torch.from_numpy(synth.make_waveforms(4, 32000, seed=72)).to(DEV)
    dm = defended_model(hip_model, defense=[(1, FeCoDefense(0.5))])
    y = dm.make_decision(x)[0]
    adv, success = PGD(dm, epsilon=0.002, step_size=0.0004, max_iter=5, batch_size=4, verbose=0).attack(x, y)
    assert (adv - x).abs().max().item() <= 0.002 + 1e-7 and adv.abs().max().item() <= 1.0
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    l0 = dm.loss_grad(x, y, SEC4SR_CrossEntropy(), want_grad=False)[2]
    l1 = dm.loss_grad(adv, y, SEC4SR_CrossEntropy(), want_grad=False)[2]
    log("PGD-5 vs FeCo-defended xv_plda: CE loss %s -> %s, success %s" % (l0.cpu().numpy().round(3), l1.cpu().numpy().round(3), success))
    assert (l1 >= l0 - 1e-4).all()  # untargeted CE ascent
    # the reference-signature function gives the same forward
    f = hip_model.compute_feat(x, flag=1)
    assert torch.equal(FeCo(f, 'kmeans', 0.5, 'L2'), FeCoDefense(0.5)(f))
    with pytest.raises(NotImplementedError):
        FeCo(f, 'warped_kmeans', 0.5, 'ts')


# ------------------------------------------------------------------------------ BASELINE config 4 in miniature
def test_audionet_feco_gradient_and_pgd_eot():
    """PGD + EOT against a FeCo-defended AudioNet (BASELINE.json configs[3], small batch): the chained gradient
    wav -> log-mel -> FeCo -> CNN vs the oracle's autograd with the device's cluster ids, then the attack itself.
    The AudioNet oracle is pinned by a run of the reference's own class (tests/golden/an_ref.npz) and the
    step after the clustering by tests/golden/feco_ref.npz; the cluster ids themselves are this repository's contract."""
    from oracle import attacks as oatk
    from oracle import feco
    from oracle.audionet import AudioNet
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    from speakerguard_amd.defense.feature_level import FeCoDefense
    from speakerguard_amd.model.audionet_csine import audionet_csine
    from speakerguard_amd.model.defended_model import defended_model
    sd = synth.make_audionet_state_dict(seed=0, num_class=251)
    hip, ora = audionet_csine.from_weights(sd, device=DEV), AudioNet(sd)
    x = torch.from_numpy(synth.make_waveforms(3, 32000, seed=73))
    d = FeCoDefense(0.5)
    dm = defended_model(hip, defense=[(1, d)])
    y = dm.make_decision(x.to(DEV))[0]
    dec, scores, loss, grad = dm.loss_grad(x.to(DEV), y, SEC4SR_CrossEntropy())
    ids = d.fwd(hip.compute_feat(x.to(DEV), flag=1))[1][0].cpu().numpy()
    xin = x.clone().requires_grad_(True)
    feats = ora.compute_feat(xin, flag=1)
    k = feats.shape[1] // 2
    same = np.mean([np.mean(feco.kmeans_ids(feats[b].detach().numpy(), k) == ids[b]) for b in range(3)])
    comp = torch.stack([feco.compress_from_ids(feats[b], ids[b], k, force=True) for b in range(3)])
    _, sc = ora.make_decision(comp, flag=1)
    oatk.cross_entropy_loss(sc, y.cpu()).backward(torch.ones(3))
    want, got = xin.grad.numpy(), grad.cpu().numpy()
    gs = np.abs(want).max()
    err = np.abs(got - want).max() / gs
    log("FeCo-defended AudioNet: ids equal %.4f, logits err %.3e, wav grad err/max %.3e" % (
        same, (scores.cpu() - sc.detach()).abs().max().item(), err))
    assert dec.cpu().tolist() == sc.argmax(1).tolist()
    bad = float((np.abs(got - want) > 3e-3 * gs).mean())
    assert bad < 5e-3 and err < 2e-2, (bad, err)
    atk = PGD(dm, epsilon=0.002, step_size=0.0004, max_iter=10, batch_size=3, EOT_size=2, EOT_batch_size=1, verbose=0)
    adv, success = atk.attack(x.to(DEV), y)
    l0 = dm.loss_grad(x.to(DEV), y, SEC4SR_CrossEntropy(), want_grad=False)[2]
    l1 = dm.loss_grad(adv, y, SEC4SR_CrossEntropy(), want_grad=False)[2]
    log("PGD-10 + EOT 2/1 vs FeCo-defended AudioNet: CE loss %s -> %s, success %s" % (
        l0.cpu().numpy().round(3), l1.cpu().numpy().round(3), success))
    assert (adv - x.to(DEV)).abs().max().item() <= 0.002 + 1e-7
    assert (l1 >= l0 - 1e-4).all()


def test_audionet_feco_fused_loop(capsys):
    """BASELINE.json configs[3] as ONE device-resident loop (sg_an_pgd_run_feco): (1) with the deterministic defense it
    is the host-chained loop of defended_model bit for bit; (2) with the randomised defense and EOT it equals a replay of
    its passes through the per-stage entry points with the same generator keys -- the repeats' gradients summed at the
    feature level in repeat order, one front-end pass and one adjoint per step;
    (3) through the PGD class it is what runs, reproducibly, and the EOT attack raises the loss of the defended model."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    from speakerguard_amd.defense.feature_level import FeCoDefense
    from speakerguard_amd.model.audionet_csine import audionet_csine
    from speakerguard_amd.model.defended_model import defended_model
    sd = synth.make_audionet_state_dict(seed=0, num_class=251)
    hip = audionet_csine.from_weights(sd, device=DEV)
    x = torch.from_numpy(synth.make_waveforms(4, 32000, seed=74)).to(DEV)
    spec = SEC4SR_CrossEntropy()
    eps, step, K = 0.002, 0.0004, 4
    lower, upper = torch.clamp(x - eps, min=-1), torch.clamp(x + eps, max=1)

    # (1) deterministic defense: fused == host-chained
    dm = defended_model(hip, defense=[(1, FeCoDefense(0.5))])
    y = dm.make_decision(x)[0]
    fused = PGD(dm, epsilon=eps, step_size=step, max_iter=K, batch_size=4, verbose=0)
    assert fused._fused_feco(4) is not None and fused._fused_feco(1) is None
    chained = PGD(dm, epsilon=eps, step_size=step, max_iter=K, batch_size=4, verbose=0)
    chained.fuse_defended = False
    adv_f, succ_f = fused.attack(x, y)
    adv_c, succ_c = chained.attack(x, y)
    assert torch.equal(adv_f, adv_c) and list(succ_f) == list(succ_c)
    # verbose traces come from the device loop too
    PGD(dm, epsilon=eps, step_size=step, max_iter=1, batch_size=4, verbose=1).attack(x, y)
    assert "iter:1" in capsys.readouterr().out

    # (2) randomised defense + EOT R: replay of the passes with the same keys (R = 3 after R = 2: odd count, workspace regrown)
    for R in (2, 3):
        feco = FeCoDefense(0.5, init='random', seed=7)
        x_adv, success, dec, scores, loss, ltr, dtr = hip.pgd_run_feco(x, y, lower, upper, spec, step, K, 1, feco, eot_size=R,
                                                                       eot_batch_size=R, trace=True)
        base_seed = hip.last_fused_seed
        replay = FeCoDefense(0.5, init='random', seed=123)  # keys are given explicitly below
        xr = x.clone()
        for it in range(K):
            feats, saved = hip.frontend_forward(xr)  # only the defense is random: one front-end pass per step
            dsum, lsum, decs = None, None, []
            for r in range(R):
                comp, sv = replay.fwd(feats, seed=hip.fused_pass_seed(base_seed, it, r))
                dec_p, _, ls_p, g = hip.loss_grad(comp, y, spec, flag=1)
                lsum = ls_p if lsum is None else lsum + ls_p
                decs.append(dec_p.cpu().tolist())
                df = replay.bwd(sv, g)
                dsum = df if dsum is None else dsum + df  # feature-level sum in repeat order (the compression is linear)
            # the per-step records are the reference's (attack/FGSM.py:50-58): loss averaged, decision voted over the repeats
            from collections import Counter
            # (numpy's true division: torch divides a device tensor by a host scalar as a multiplication by 1 / R)
            assert np.array_equal(lsum.cpu().numpy() / np.float32(R), ltr[it].cpu().numpy()), (R, it)
            assert dtr[it].cpu().tolist() == [Counter(decs[r][b] for r in range(R)).most_common(1)[0][0] for b in range(x.shape[0])]
            gw = hip.frontend_backward(saved, dsum)
            hip.pgd_update(xr, gw.contiguous(), lower.contiguous(), upper.contiguous(), step, 1)
        comp, _ = replay.fwd(hip.compute_feat(xr, flag=1), seed=hip.fused_pass_seed(base_seed, K, 0))
        dec_r, sc_r = hip.make_decision(comp, flag=1)
        assert torch.equal(xr, x_adv) and torch.equal(dec_r, dec) and torch.equal(sc_r, scores), R
        assert torch.equal(ltr[K], loss) and torch.equal(dtr[K], dec)
        assert success.bool().tolist() == (dec != y).tolist()
        assert not torch.equal(x_adv, adv_f)  # the random clusterings lead somewhere else than the evenly started one

    # (3) the PGD class against the randomised defense: device loop, reproducible, loss goes up
    def run(seed):
        d = FeCoDefense(0.5, init='random', seed=seed)
        m = defended_model(hip, defense=[(1, d)])
        a = PGD(m, epsilon=eps, step_size=step, max_iter=10, batch_size=4, EOT_size=4, EOT_batch_size=2, verbose=0)
        hip._noise_epoch = 0
        adv, succ = a.attack(x, y)
        return adv, succ, d.calls
    a1, s1, calls = run(5)
    a2, s2, _ = run(5)
    a3, _, _ = run(6)
    assert calls == 1, "one fused call per batch"
    assert torch.equal(a1, a2) and list(s1) == list(s2) and not torch.equal(a1, a3)
    assert (a1 - x).abs().max().item() <= eps + 1e-7
    det = defended_model(hip, defense=[(1, FeCoDefense(0.5))])
    l0 = det.loss_grad(x, y, spec, want_grad=False)[2]
    l1 = det.loss_grad(a1, y, spec, want_grad=False)[2]
    log("fused PGD + EOT vs FeCo-defended AudioNet: == host-chained loop (deterministic defense), == keyed replay (random "
        "init, EOT 2); PGD-10 EOT 4/2: CE loss %s -> %s, success %s" % (l0.cpu().numpy().round(3), l1.cpu().numpy().round(3), s1))
    assert (l1 >= l0 - 1e-4).all()


def test_score_vjp_and_average_order_gradient(hip_model, oracle_model):
    """(1) SG_LOSS_LINEAR: d(sum coef * scores)/d x for an arbitrary coef -- what a caller-defined loss of the scores needs
    (the reference gets it from autograd, EOT.py:33-35) -- vs the oracle's autograd, x-vector and AudioNet.
    (2) model/defended_model.py 'average' order: loss of the MEAN score over three defended branches (quantisation at the
    waveform behind the reference's straight-through BPDA, FeCo at feature levels 1 and 2) vs the oracle's autograd
    through the same composition with the device's cluster ids."""
    from oracle import attacks as oatk
    from oracle import feco
    from oracle.audionet import AudioNet
    from speakerguard_amd import synth
    from speakerguard_amd.adaptive_attack.BPDA import straight_through
    from speakerguard_amd.attack.utils import ScoreVJP, SEC4SR_CrossEntropy, SEC4SR_MarginLoss, loss_dscores
    from speakerguard_amd.defense.feature_level import FeCoDefense
    from speakerguard_amd.model.audionet_csine import audionet_csine
    from speakerguard_amd.model.defended_model import defended_model
    rs = np.random.RandomState(12)
    x = torch.from_numpy(synth.make_waveforms(3, 32000, seed=75))
    y0 = torch.zeros(3, dtype=torch.int64, device=DEV)

    def rel(got, want):
        return float(np.abs(got - want).max() / np.abs(want).max())

    # (1) vector-Jacobian product of the scores
    coef = torch.from_numpy(rs.randn(3, 10).astype(np.float32))
    dec, sc, ls, g = hip_model.loss_grad(x.to(DEV), y0, ScoreVJP(coef.to(DEV)))
    xin = x.clone().requires_grad_(True)
    osc = oracle_model.make_decision(xin)[1]
    (coef * osc).sum().backward()
    e_xv = rel(g.cpu().numpy(), xin.grad.numpy())
    # A random +-coef over all ten scores makes the ten score gradients cancel, so the two fp32 implementations sit further
    # apart (relative to the largest entry) than under a loss; both are judged against the SAME model evaluated in fp64:
    # the HIP gradient must be about as close to it as the fp32 oracle is (measured 8e-3 vs 6e-3 rms; under the
    # cross-entropy loss, test_waveform_gradient_matches_oracle_autograd, 2e-4 vs 1e-3).
    from oracle.xv_plda import XvPlda
    m64 = XvPlda(synth.make_xv_weights()).double()
    x64 = x.double().requires_grad_(True)
    (coef.double() * m64.make_decision(x64)[1]).sum().backward()
    g64 = x64.grad.numpy()
    rms = lambda a: float(np.sqrt(((a - g64) ** 2).mean() / (g64 ** 2).mean()))
    rms_hip, rms_ora = rms(g.cpu().numpy()), rms(xin.grad.numpy())
    np.testing.assert_allclose(ls.cpu().numpy(), (coef * osc.detach()).sum(1).numpy(), rtol=2e-4, atol=2e-2)
    sd = synth.make_audionet_state_dict(seed=0, num_class=251)
    an, oan = audionet_csine.from_weights(sd, device=DEV), AudioNet(sd)
    coef_a = torch.from_numpy(rs.randn(3, 251).astype(np.float32))
    _, _, _, ga = an.loss_grad(x.to(DEV), y0, ScoreVJP(coef_a.to(DEV)))
    xin = x.clone().requires_grad_(True)
    (coef_a * oan.make_decision(xin)[1]).sum().backward()
    e_an = rel(ga.cpu().numpy(), xin.grad.numpy())
    assert rms_hip <= 2 * rms_ora and rms_hip < 2e-2 and e_xv < 5e-2 and e_an < 2e-3, (rms_hip, rms_ora, e_xv, e_an)
    with pytest.raises(ValueError, match="ScoreVJP"):  # a (B, S) table of the wrong shape is refused before the pass reads it
        hip_model.loss_grad(x.to(DEV), y0, ScoreVJP(coef[:2].to(DEV)))
    # the loss stage alone on given scores == the tail kernel's own
    spec = SEC4SR_MarginLoss(targeted=False, confidence=0.5, task='CSI', threshold=None, clip_max=True)
    yv = torch.tensor([1, 7, 3], device=DEV)
    d1, s1, l1, _ = hip_model.loss_grad(x.to(DEV), yv, spec, want_grad=False)
    d2, l2, ds2 = loss_dscores(hip_model, s1, yv, spec)
    assert torch.equal(d1, d2) and torch.equal(l1, l2) and ds2.shape == s1.shape

    # (2) 'average' order
    quant = lambda w: torch.round(w * 512.0) / 512.0
    f1, f2 = FeCoDefense(0.5), FeCoDefense(0.4)
    dm = defended_model(hip_model, defense=[(0, straight_through(quant)), (1, f1), (2, f2)], order='average')
    with torch.no_grad():
        y = oracle_model.make_decision(x)[0]
    ce = SEC4SR_CrossEntropy()
    dec, mean, loss, grad = dm.loss_grad(x.to(DEV), y.to(DEV), ce)
    assert torch.equal(mean, dm.score(x.to(DEV)))  # the forward-only path of the same wrapper
    ids1 = f1.fwd(hip_model.compute_feat(x.to(DEV), flag=1))[1][0].cpu().numpy()
    ids2 = f2.fwd(hip_model.compute_feat(x.to(DEV), flag=2))[1][0].cpu().numpy()
    xin = x.clone().requires_grad_(True)
    s0 = oracle_model.make_decision(xin + (quant(xin) - xin).detach())[1]  # straight-through quantisation
    fe1 = oracle_model.compute_feat(xin, flag=1)
    k1 = int(fe1.shape[1] * 0.5)
    s1 = oracle_model.make_decision(torch.stack([feco.compress_from_ids(fe1[b], ids1[b], k1, force=True) for b in range(3)]), flag=1)[1]
    fe2 = oracle_model.compute_feat(xin, flag=2)
    k2 = int(fe2.shape[1] * 0.4)
    s2 = oracle_model.make_decision(torch.stack([feco.compress_from_ids(fe2[b], ids2[b], k2, force=True) for b in range(3)]), flag=2)[1]
    om = (s0 + s1 + s2) / 3
    ol = oatk.cross_entropy_loss(om, y)
    ol.backward(torch.ones(3))
    want, got = xin.grad.numpy(), grad.cpu().numpy()
    err = rel(got, want)
    bad = float((np.abs(got - want) > 3e-3 * np.abs(want).max()).mean())
    log("score VJP: xv rms error vs fp64 truth %.2e (fp32 oracle: %.2e), max |hip - oracle| %.2e of max; AudioNet %.2e of max; 'average' order over (BPDA quantise, FeCo@1, FeCo@2): "
        "mean-score err %.2e, loss err %.2e, wav grad err/max %.2e (outliers %.4f)" % (
            rms_hip, rms_ora, e_xv, e_an, (mean.cpu() - om.detach()).abs().max().item(), (loss.cpu() - ol.detach()).abs().max().item(), err, bad))
    assert dec.cpu().tolist() == om.argmax(1).tolist()
    assert (mean.cpu() - om.detach()).abs().max().item() < 5e-3
    assert bad < 5e-3 and err < 2e-2, (bad, err)


def test_average_order_uses_one_dither_realisation_per_branch(xv_weights):
    """with the reference's default dither (xv_plda.py:119 dither = 1.0) a waveform-level branch of the 'average'
    order scored its input with one noise draw and back-propagated through ANOTHER (two native passes, two draws); the
    reference's autograd graph holds one.  Now one key serves both: the returned mean score and the returned gradient are
    those of the same realisation (replayed here with the key), and a branch costs one draw."""
    from speakerguard_amd import synth
    from speakerguard_amd.adaptive_attack.BPDA import straight_through
    from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
    from speakerguard_amd.model.defended_model import defended_model
    from speakerguard_amd.model.xv_plda import xv_plda
    m = xv_plda.from_weights(xv_weights, device=DEV, dither=1.0, dither_seed=5)
    x = torch.from_numpy(synth.make_waveforms(3, 24000, seed=81)).to(DEV)
    y = torch.tensor([1, 4, 7], device=DEV)
    quant = lambda w: torch.round(w * 512.0) / 512.0
    dm = defended_model(m, defense=[(0, straight_through(quant))], order='average')
    ce = SEC4SR_CrossEntropy()
    draw0 = m._draw
    key = m.noise_seed(m.dither_seed, draw0)
    dec, mean, loss, grad = dm.loss_grad(x, y, ce)
    assert m._draw == draw0 + 1                                     # one draw for the branch, not two
    xq = quant(x)
    assert torch.equal(mean, m.forward(xq, flag=0, dither_seed=key))  # the scores of THAT realisation
    d2, s2, l2, g2 = m.loss_grad(xq, y, ce, dither_seed=key)
    assert torch.equal(dec, d2) and torch.allclose(loss, l2, rtol=1e-5, atol=1e-6)
    scale = g2.abs().max().item()
    assert (grad - g2).abs().max().item() < 2e-4 * scale            # ... and its gradient (score-VJP vs direct CE chain)
    g_other = m.loss_grad(xq, y, ce, dither_seed=key + 1)[3]          # another realisation is measurably different
    assert (g_other - g2).abs().max().item() > 50 * (grad - g2).abs().max().item()


def test_randomised_feco_is_shard_invariant():
    """The random initial frames of the defense are keyed like the dither -- (seed, attack call, restart, GLOBAL index of the
    chunk's first utterance, call number inside the chunk) + the row inside the chunk -- so the two halves of a batch
    attacked as two ranks would (attacker.index_offset = shard start) reproduce the unsharded run bit for bit: in the
    device loop and in the host-chained loop."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.defense.feature_level import FeCoDefense
    from speakerguard_amd.model.audionet_csine import audionet_csine
    from speakerguard_amd.model.defended_model import defended_model
    hip = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=DEV)
    x = torch.from_numpy(synth.make_waveforms(4, 24000, seed=76)).to(DEV)
    y = hip.make_decision(x)[0]
    for fused in (True, False):
        def make():
            dm = defended_model(hip, defense=[(1, FeCoDefense(0.5, init='random', seed=3))])
            a = PGD(dm, epsilon=0.002, step_size=0.0005, max_iter=3, batch_size=2, EOT_size=2, EOT_batch_size=2, verbose=0)
            a.fuse_defended = fused
            return a
        hip._noise_epoch = 0
        full = make().attack(x, y)
        parts = []
        for lo, hi in ((0, 2), (2, 4)):
            hip._noise_epoch = 0
            a = make()
            a.index_offset = lo
            parts.append(a.attack(x[lo:hi], y[lo:hi]))
        assert torch.equal(full[0], torch.cat([p[0] for p in parts], 0)), fused
        assert list(full[1]) == sum((list(p[1]) for p in parts), [])
        hip._noise_epoch = 0
        shifted = make()
        shifted.index_offset = 2  # utterances 0, 1 attacked as "global utterances 2, 3" see other clusterings
        assert not torch.equal(shifted.attack(x[0:2], y[0:2])[0], full[0][0:2])
    log("randomised FeCo (device loop and host-chained loop): halves == full batch bit for bit")


def test_sharding_a_feco_defended_model_keeps_single_utterance_calls_where_they_were():
    """FeCo is batch-coupled -- a model call with ONE utterance drops empty clusters, a larger one fills
    them in (reference defense/feature_level.py:33).  ShardedAttack therefore never turns an utterance of a multi-utterance
    call into a call of its own: N = world + 1 = 9 utterances on 8 ranks are 3 + 2 + 2 + 2 (four ranks idle), N = 65 with
    batch_size 64 keeps the unsharded run's trailing one-utterance call on the last rank; the ranks' results put together
    are the single-GPU result bit for bit."""
    from speakerguard_amd import synth
    from speakerguard_amd.attack.PGD import PGD
    from speakerguard_amd.defense.feature_level import FeCoDefense
    from speakerguard_amd.model.audionet_csine import audionet_csine
    from speakerguard_amd.model.defended_model import defended_model
    from speakerguard_amd.shard import ShardedAttack
    hip = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=DEV)

    class OneRank(ShardedAttack):
        def __init__(self, attacker, world, rank):
            super().__init__(attacker, gather_audio=True)
            self.w, self.r = world, rank

        def _world(self):
            return self.w, self.r

        def _gather_rows(self, local, bounds, n):
            out = torch.zeros((n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            s, e = bounds[self.r]
            out[s:e] = local[: e - s]
            return out

    for n, bs, world in ((9, 64, 8), (65, 64, 8), (5, 2, 2)):
        x = torch.from_numpy(synth.make_waveforms(n, 16000, seed=300 + n)).to(DEV)
        y = hip.make_decision(x)[0]

        def make():
            dm = defended_model(hip, defense=[(1, FeCoDefense(0.5, init='random', seed=3))])
            return PGD(dm, epsilon=0.002, step_size=0.0005, max_iter=2, batch_size=bs, EOT_size=2, EOT_batch_size=2, verbose=0)
        hip._noise_epoch = 0
        ref_adv, ref_succ = make().attack(x, y)
        adv, flags, sizes = torch.zeros_like(ref_adv), [False] * n, []
        for rank in range(world):
            hip._noise_epoch = 0
            atk = make()
            calls = []
            inner = atk.attack_batch
            atk.attack_batch = lambda xb, *a, inner=inner, calls=calls: (calls.append(int(xb.shape[0])), inner(xb, *a))[1]
            a, f = OneRank(atk, world, rank).attack(x, y)
            adv += a
            flags = [p or bool(q) for p, q in zip(flags, f)]
            sizes.append(calls)
        assert torch.equal(adv, ref_adv) and flags == [bool(v) for v in ref_succ], (n, bs, world, sizes)
        log("FeCo-defended AudioNet, %d utterances, batch_size %d, %d emulated ranks: model calls per rank %s == the single-GPU attack bit for bit"
            % (n, bs, world, sizes))
