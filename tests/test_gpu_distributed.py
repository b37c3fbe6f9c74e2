"""GPU, two ranks: the product's own data-parallel step -- FusedTrainStep.step() with world_size 2, i.e. the HIP forward /
backward, the world > 1 branch (ONE all-reduce of [gradients | EMA statistics]), 1/R inside the Adam kernel, and the
replication of rank 0's state at construction -- against the reference-generated DP fixture (tests/golden/model_tiny.npz
dp.*: R replicas with per-rank BatchNorm, averaged gradients, one Adam step).

The two ranks are fresh child processes sharing GPU 0 over gloo (NSG_DIST_BACKEND=gloo NSG_DEVICE_INDEX=0); they are
started by tests/conftest.py before the pytest process touches the GPU (tests/helpers/dp_rank.py is what each runs).
RCCL itself only exists on a multi-GPU node: that leg is the driver's `bench.py --gpus N`."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from neural_sound_generation_amd import models as M  # noqa: E402
from neural_sound_generation_amd.train import FusedTrainStep  # noqa: E402

DEV = "cuda:0"
BIAS_BEFORE_BN = ("encoder.0.bias", "block.1.bias", "block.4.bias", "decoder.3.bias")   # exactly-zero gradients: round-off noise


@pytest.fixture(scope="module")
def ranks(dp_rehearsal_dir):
    assert dp_rehearsal_dir, "conftest did not run the two-rank rehearsal (needs -m gpu and a visible GPU)"
    rcs = open(os.path.join(dp_rehearsal_dir, "rcs.txt")).read().split()
    logs = "\n".join("---- rank %d ----\n%s" % (r, open(os.path.join(dp_rehearsal_dir, "rank%d.log" % r)).read()[-4000:]) for r in range(2))
    assert rcs == ["0", "0"], "a rank failed:\n" + logs
    return [np.load(os.path.join(dp_rehearsal_dir, "rank%d.npz" % r)) for r in range(2)]


def sub(npz, prefix):
    return {k[len(prefix) + 1:]: npz[k] for k in npz.files if k.startswith(prefix + "/")}


def test_state_is_replicated_from_rank0_at_construction(ranks, golden_dir):
    """Rank 1 was built from another seed with perturbed BatchNorm buffers: after FusedTrainStep.__init__ every tensor of
    its state_dict -- parameters (the flat bucket) AND buffers (running statistics, counters) -- equals rank 0's."""
    g = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    a0, a1 = sub(ranks[0], "after_init.0"), sub(ranks[1], "after_init.1")
    assert set(a0) == set(a1) and len(a0) > 40
    for k in a0:
        assert np.array_equal(a0[k], a1[k]), k
        assert np.array_equal(a0[k], g["sd0." + k]), k


def test_two_rank_step_matches_reference_dp_fixture(ranks, golden_dir):
    g = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    for r in range(2):      # per-rank losses (per-rank BatchNorm statistics)
        np.testing.assert_allclose(ranks[r]["losses"], g["dp.losses"][r], rtol=1e-5)
    grad0, grad1 = sub(ranks[0], "grad"), sub(ranks[1], "grad")
    sd0, sd1 = sub(ranks[0], "sd1"), sub(ranks[1], "sd1")
    for k in grad0:
        assert np.array_equal(grad0[k], grad1[k]), f"{k}: the all-reduced gradient differs between ranks"
        assert np.array_equal(sd0[k], sd1[k]), f"{k}: the ranks took different steps"
        if k.endswith(BIAS_BEFORE_BN):
            wscale = float(np.abs(grad0[k[:-4] + "weight"]).max())
            assert float(np.abs(grad0[k]).max()) <= 1e-3 * wscale + 1e-6, k
            continue
        want = g["dp.grad." + k]
        scale = max(float(np.abs(want).max()), 1e-8)
        assert float(np.abs(grad0[k] - want).max()) <= 2e-4 * scale + 1e-8, k
        big = np.abs(want) > 1e-5
        np.testing.assert_allclose(sd0[k][big], g["dp.sd1." + k][big], rtol=0, atol=5e-6, err_msg=k)
    # BatchNorm statistics stay per rank (plain DDP semantics): the ranks saw different clips
    assert not np.array_equal(sd0["encoder.1.running_mean"], sd1["encoder.1.running_mean"])
    assert int(sd0["encoder.1.num_batches_tracked"]) == int(sd1["encoder.1.num_batches_tracked"]) == int(g["sd0.encoder.1.num_batches_tracked"]) + 1


def test_two_rank_ema_statistics_ride_in_the_gradient_all_reduce(ranks, golden_dir):
    """EMA codebook mode (extension): per-code counts / sums are summed over ranks by the SAME collective as the gradients
    and every rank applies the identical codebook update.  Expected values: the two shards run here in one process."""
    g = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    e0, e1 = sub(ranks[0], "ema_after_init"), sub(ranks[1], "ema_after_init")
    for k in e0:        # rank 1's codebook / ema_count / ema_sum were perturbed: the constructor replicates rank 0's
        assert np.array_equal(e0[k], e1[k]), k
    for k in ("ema_n", "ema_s", "ema_codebook", "ema_count"):
        assert np.array_equal(ranks[0][k], ranks[1][k]), k
    n_sum = s_sum = None
    for r in range(2):
        torch.manual_seed(1)
        m = M.VQVAE(1, 16, 32, ema_decay=0.99).to(DEV).train()
        st = FusedTrainStep(m, lr=1e-3)
        st.forward_backward(torch.from_numpy(g["dp.c%d" % r]).to(DEV))
        n_sum = st.ema_n.clone() if n_sum is None else n_sum + st.ema_n
        s_sum = st.ema_s.clone() if s_sum is None else s_sum + st.ema_s
    assert np.array_equal(ranks[0]["ema_n"], n_sum.cpu().numpy())            # counts are exact integers
    assert float(n_sum.sum().item()) == 2 * 2 * 20 * 16                        # every latent row of both shards counted once
    np.testing.assert_allclose(ranks[0]["ema_s"], s_sum.cpu().numpy(), rtol=1e-5, atol=1e-6)
    torch.manual_seed(1)
    m = M.VQVAE(1, 16, 32, ema_decay=0.99).to(DEV).train()
    st = FusedTrainStep(m, lr=1e-3)
    st.apply_ema(n_sum, s_sum)
    np.testing.assert_allclose(ranks[0]["ema_codebook"], m.codebook.embedding.weight.detach().cpu().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ranks[0]["ema_count"], m.codebook.ema_count.cpu().numpy(), rtol=1e-6)
