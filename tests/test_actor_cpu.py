"""CPU: the device actor's host side and its oracle (RL/MR_ddpg.py:80-160 ActorNetwork, :59-78 OUNoise).

Parity with the reference's TF1 / tflearn network is UNPINNED (neither library is installable here).  What is pinned:
the oracle's restatement (oracle/mrsim_oracle.c: orc_actor_forward, the fp32 summation order the MFMA kernel has) against the
PyTorch twin's fp32 forward, the library's batch-norm folding and packing (host entry points of libmrsim.so: no GPU
needed), the specified tanh against libm, and the OU recurrence against its closed-form moments."""
import ctypes as C
import math

import numpy as np
import torch

from mr_rl_amd import _lib
from mr_rl_amd.actor import fold_actor, fold_bn, pack_weights
from oracle import oracle as O
from tests.util import random_actor

ACTOR_TOL = 1e-5   # relative to action_bound: HIP / oracle fp32 (fmaf chains) vs the PyTorch fp32 forward


def _obs(n, seed=0):
    r = np.random.default_rng(seed)
    o = np.zeros((n, 5), dtype=np.float32)
    o[:, :2] = r.uniform(-300, 300, (n, 2))
    o[:, 2:4] = r.uniform(-50, 50, (n, 2))
    o[:, 4] = np.hypot(o[:, 2] - o[:, 0], o[:, 3] - o[:, 1])
    return o


def test_fold_bn_matches_eval_mode_batchnorm():
    m = random_actor(3)
    x = torch.randn(257, 5)
    w1, b1 = fold_bn(m.fc1.weight.detach().numpy(), m.fc1.bias.detach().numpy(), m.bn1.weight.detach().numpy(),
                     m.bn1.bias.detach().numpy(), m.bn1.running_mean.numpy(), m.bn1.running_var.numpy(), m.bn1.eps)
    want = m.bn1(m.fc1(x)).detach().numpy()
    got = x.numpy() @ w1.T + b1
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)


def test_pack_layout_places_every_weight_where_the_kernel_reads_it():
    """mrsim_actor_pack_host against the layout mrsim_actor.h documents: A operand of lane l at k-step s = W[32 rt + (l & 31)]
    [k], k = 2 s + (l >> 5) for layer 1 and kperm(s, l >> 5) for layer 2; bias / output weights per lane half in
    accumulator-register order."""
    H = 64
    kperm = lambda q, h: 32 * (q // 16) + 8 * ((q % 16) // 4) + 4 * h + (q % 4)  # noqa: E731
    assert sorted(kperm(q, h) for q in range(32) for h in range(2)) == list(range(64))
    w = {"w1": np.arange(H * 5, dtype=np.float32).reshape(H, 5) + 1, "b1": 1000 + np.arange(H, dtype=np.float32),
         "w2": 2000 + np.arange(H * H, dtype=np.float32).reshape(H, H), "b2": 7000 + np.arange(H, dtype=np.float32),
         "w3": 8000 + np.arange(2 * H, dtype=np.float32).reshape(2, H), "b3": np.float32([9001, 9002]),
         "obs_scale": np.float32([1, 2, 4, 8, 16]), "action_bound": np.float32([20, 6.25])}
    blob = pack_weights(w)
    assert blob.shape == (_lib.ACTOR_BLOB_FLOATS,)
    A1 = blob[:384].reshape(2, 3, 64)
    A2 = blob[384:384 + 4096].reshape(2, 8, 64, 4)
    C1, C2 = blob[4480:4544].reshape(2, 32), blob[4544:4608].reshape(2, 32)
    W3 = blob[4608:4736].reshape(2, 2, 32)
    for rt in range(2):
        for lane in range(64):
            f, h = 32 * rt + (lane & 31), lane >> 5
            for s in range(3):
                k = 2 * s + h
                assert A1[rt, s, lane] == (w["w1"][f, k] * w["obs_scale"][k] if k < 5 else 0.0)
            for q in range(32):
                assert A2[rt, q // 4, lane, q % 4] == w["w2"][f, kperm(q, h)]
    for h in range(2):
        for q in range(32):
            assert C1[h, q] == w["b1"][kperm(q, h)] and C2[h, q] == w["b2"][kperm(q, h)]
            for o in range(2):
                assert W3[h, o, q] == w["w3"][o, kperm(q, h)]
    assert list(blob[4736:4740]) == [9001, 9002, 20, 6.25] and not blob[4740:4744].any()
    # bf16x3 section [rt 2][s 4][part 3][lane 64][8 bf16]: the three terms of W2[f][kperm(8 s + jj, h)] sum back to it exactly
    bf = blob[4744:].view(np.uint16).reshape(2, 4, 3, 64, 8)
    as_f32 = (bf.astype(np.uint32) << 16).view(np.float32)
    for rt in range(2):
        for lane in (0, 17, 31, 32, 63):
            f, h = 32 * rt + (lane & 31), lane >> 5
            for sk in range(4):
                for jj in range(8):
                    terms = as_f32[rt, sk, :, lane, jj].astype(np.float64)
                    assert terms.sum() == float(w["w2"][f, kperm(8 * sk + jj, h)])
    bad = _lib.MrsimActorWeights()
    assert _lib.lib().mrsim_actor_pack_host(C.byref(bad), blob.ctypes.data_as(C.c_void_p)) == _lib.EINVAL


def test_spec_tanh_against_libm():
    L = O.lib()
    xs = np.concatenate([np.linspace(-12, 12, 20001), np.linspace(-0.7, 0.7, 20001), [0.0, 0.625, -0.625, 9.0, -9.0, 1e-8]])
    worst = max(abs(L.orc_spec_tanhf(float(np.float32(x))) - math.tanh(float(np.float32(x)))) for x in xs)
    assert worst < 2e-7, worst
    assert L.orc_spec_tanhf(0.0) == 0.0 and L.orc_spec_tanhf(20.0) == 1.0 and L.orc_spec_tanhf(-20.0) == -1.0


def test_oracle_actor_matches_the_pytorch_twin():
    """orc_actor_forward (fp32 fmaf chains in the MFMA kernel's order, folded batch norm, specified tanh) vs
    mr_rl_amd.ddpg.Actor.forward in eval mode, raw and scaled observations, small and saturating output layers."""
    for seed, scale, out_scale in ((0, None, None), (1, [0.01] * 5, None), (2, [0.01] * 5, 40.0), (3, None, 2.0)):
        m = random_actor(seed, out_scale=out_scale)
        obs = _obs(4096, seed)
        sc = torch.ones(5) if scale is None else torch.tensor(scale)
        with torch.no_grad():
            want = m(torch.from_numpy(obs) * sc).numpy()
            m64 = random_actor(seed, out_scale=out_scale).double()
            want64 = m64(torch.from_numpy(obs).double() * sc.double()).numpy()
        got = O.actor_forward(O.make_actor(fold_actor(m, scale)), obs)
        bound = m.action_bound.numpy()
        err = np.abs(got - want) / bound
        # what fp32 evaluation order alone is worth on this input: the twin's own fp32 forward against its fp64 forward
        # (raw observations of a few hundred units drive the pre-activations to ~1e2, where an ulp is 1e-5 of the bound)
        cond = (np.abs(want - want64) / bound).max()
        assert err.max() < max(ACTOR_TOL, 3 * cond), (seed, err.max(), cond)
        assert (np.abs(got - want64) / bound).max() < max(ACTOR_TOL, 3 * cond)
        if out_scale == 40.0:
            assert (np.abs(want[:, 0]) > 0.99 * bound[0]).any()      # tanh saturates somewhere
        assert np.abs(want).max() > 1e-3


def test_oracle_ou_recurrence_and_moments():
    """OUNoise.__call__ (RL/MR_ddpg.py:69-73), mu = 0: x' = x - theta x dt + sigma sqrt(dt) N.  With a zero network the
    action IS the OU state: check the recurrence on the generator's own normals and the variance after t steps."""
    m = random_actor(0)
    with torch.no_grad():
        for p in (m.out.weight, m.out.bias):
            p.zero_()
    n, steps, seed = 8192, 150, 11
    A = O.make_actor(fold_actor(m), ou=True)
    ou = np.zeros((n, 2), dtype=np.float32)
    obs = _obs(n)
    prev = ou.copy()
    for t in range(steps):
        act = O.actor_policy(A, obs, ou, seed, t + 1, threads=8)
        if t == 0:
            z = np.array([O.normals4(seed, i, 1, O.c0(O.STREAM_DYN))[:2] for i in range(16)])
            want = np.float32(0.3 * math.sqrt(1e-2)) * z
            np.testing.assert_allclose(ou[:16], want, rtol=1e-6, atol=0)
        assert np.array_equal(act, ou)               # tanh(0) * bound + x
        prev = ou.copy()
    th, sg, dt = 0.15, 0.3, 1e-2
    a = 1 - th * dt
    var = sg * sg * dt * (1 - a ** (2 * steps)) / (1 - a * a)
    assert abs(ou.mean()) < 4e-3 and abs(ou.std() / math.sqrt(var) - 1) < 0.03
    assert not np.array_equal(prev, np.zeros_like(prev))
    # reset_on_done: the state is zeroed before the draw of an episode's first step (counter == 0)
    A2 = O.make_actor(fold_actor(m), ou=True, reset_on_done=True)
    cnt = np.ones(n, dtype=np.int32); cnt[:100] = 0
    before = ou.copy()
    O.actor_policy(A2, obs, ou, seed, 999, counter=cnt)
    z0 = np.array([O.normals4(seed, i, 999, O.c0(O.STREAM_DYN))[:2] for i in range(4)])
    np.testing.assert_allclose(ou[:4], np.float32(0.3 * math.sqrt(1e-2)) * z0, rtol=1e-6)
    assert np.abs(ou[100:] - before[100:] * np.float32(a)).max() < 0.2   # the others continued from their state


def _bf16(x):
    """float32 -> nearest-even bf16, kept as float32 (what v_cvt_pk_bf16_f32 returns, widened again)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def test_oracle_bf16_arithmetics_against_a_numpy_emulation():
    """MrsimActor.math = BF16X3 / BF16 in the oracle (orc_actor_forward's second branch) against the same arithmetic written
    independently with numpy: operands rounded to bf16 (three-term split or one term), each 16-wide k-step summed exactly
    (fp64 holds a sum of 16 bf16 x bf16 products without rounding at these magnitudes), then ONE fp32 rounding into the
    accumulator, k-steps and terms in the kernel's order.  Also: bf16x3 stays within BF_TOL of the f32 arithmetic and plain
    bf16 within BF16_TOL, the bounds tests/test_gpu_actor.py uses for the kernel."""
    kperm = lambda q, h: 32 * (q // 16) + 8 * ((q % 16) // 4) + 4 * h + (q % 4)  # noqa: E731
    L = O.lib()
    m = random_actor(5, out_scale=20.0)
    w = fold_actor(m, [0.01] * 5)
    obs = _obs(512, 9)
    w1 = (w["w1"] * np.float32(w["obs_scale"])[None, :]).astype(np.float32)
    h1 = np.tile(w["b1"].astype(np.float32), (len(obs), 1))
    for k in range(5):                                        # fmaf chain in k order: one rounding per term
        h1 = (h1.astype(np.float64) + w1[:, k].astype(np.float64)[None, :] * obs[:, k].astype(np.float64)[:, None]).astype(np.float32)
    h1 = np.maximum(h1, 0)

    def split(x, nterm):
        t, r = [], x.astype(np.float32)
        for _ in range(nterm):
            t.append(_bf16(r))
            r = (r - t[-1]).astype(np.float32)
        return t

    # plain bf16: layer 1 on the bf16 matrix cores too -- bf16-rounded weights against the inputs' exact three-term splits,
    # all 15 products and the bias in ONE instruction: summed exactly, rounded once
    xs = split(obs, 3)
    assert np.array_equal((xs[0].astype(np.float64) + xs[1]) + xs[2], obs.astype(np.float64))      # the split loses nothing
    h1_bf = np.maximum((w["b1"].astype(np.float64)[None, :] + obs.astype(np.float64) @ _bf16(w1).astype(np.float64).T)
                       .astype(np.float32), 0)
    # bf16 x 3: layer 1 as two instructions per tile -- w1 x3 + w2 x2 + w3 x1 of all five inputs first, then w1 x1 + w1 x2 + w2 x1
    wt = split(w1, 3)
    f64 = lambda a: a.astype(np.float64)  # noqa: E731
    small = f64(xs[2]) @ f64(wt[0]).T + f64(xs[1]) @ f64(wt[1]).T + f64(xs[0]) @ f64(wt[2]).T
    big = f64(xs[0]) @ f64(wt[0]).T + f64(xs[1]) @ f64(wt[0]).T + f64(xs[0]) @ f64(wt[1]).T
    acc = (f64(w["b1"])[None, :] + small).astype(np.float32)
    h1_x3 = np.maximum((f64(acc) + big).astype(np.float32), 0)
    want = {}
    for math_name, pairs in (("bf16x3", [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]), ("bf16", [(0, 0)])):
        ws, hs = split(w["w2"], 3), split(h1_bf if math_name == "bf16" else h1_x3, 3)
        acc = np.tile(w["b2"].astype(np.float32), (len(obs), 1))
        for s in range(4):
            ks = [kperm(8 * s + jj, h) for h in range(2) for jj in range(8)]
            for (a, b) in pairs:
                part = hs[b][:, ks].astype(np.float64) @ ws[a][:, ks].astype(np.float64).T
                acc = (acc.astype(np.float64) + part).astype(np.float32)
        h2 = np.maximum(acc, 0)
        out = np.zeros((len(obs), 2), np.float32)
        for o in range(2):
            if math_name == "bf16":
                # plain bf16: the output layer on the bf16 matrix cores as well -- W3 and relu(h2) rounded once, four k-steps of 16
                # features, each summed exactly and rounded once into the f32 accumulator (which starts at 0); the bias afterwards
                a3, w3b, h2b = np.zeros(len(obs), np.float32), _bf16(w["w3"][o].astype(np.float32)), _bf16(h2)
                for s in range(4):
                    ks = [kperm(8 * s + jj, h) for h in range(2) for jj in range(8)]
                    a3 = (a3.astype(np.float64) + h2b[:, ks].astype(np.float64) @ w3b[ks].astype(np.float64)).astype(np.float32)
                pre = (a3 + np.float32(0)) + np.float32(w["b3"][o])
                out[:, o] = [np.float32(L.orc_spec_tanhf(float(x))) * np.float32(w["action_bound"][o]) for x in pre]
                continue
            p = np.zeros((len(obs), 2), np.float32)
            for h in range(2):
                for q in range(32):
                    k = kperm(q, h)
                    p[:, h] = (p[:, h].astype(np.float64) + np.float64(w["w3"][o, k]) * h2[:, k].astype(np.float64)).astype(np.float32)
            pre = (p[:, 0] + p[:, 1]) + np.float32(w["b3"][o])
            out[:, o] = [np.float32(L.orc_spec_tanhf(float(x))) * np.float32(w["action_bound"][o]) for x in pre]
        want[math_name] = out
        got = O.actor_forward(O.make_actor(w, math=math_name), obs)
        # the k-step sums of 16 products are exact in fp64 only while their exponents span < 53 - 16 bits; they do here
        np.testing.assert_array_equal(got, want[math_name], err_msg=math_name)
    f32 = O.actor_forward(O.make_actor(w), obs)
    bound = np.float32(w["action_bound"])
    assert (np.abs(want["bf16x3"] - f32) / bound).max() < 5e-6
    e1 = (np.abs(want["bf16"] - f32) / bound).max()
    assert 1e-5 < e1 < 6e-2, e1                               # plain bf16 is visibly, boundedly different
