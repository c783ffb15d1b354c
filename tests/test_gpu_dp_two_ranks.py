"""GPU, TWO ranks (gloo group, both processes on cuda:0 -- the box has one GPU, and RCCL refuses two ranks on one device): the
data-parallel path of codec_train_iteration on real kernels with each rank holding half of a global batch of four, against ONE process
on the whole batch.  What has to hold: after the all-reduce the scaled gradients are the global-batch gradients (every
term is a batch mean; the MR-STFT spectral-convergence ratio through dp.stft_sum_hook), and both ranks end with bit-identical
parameters (same gradients, same clip coefficient, same AdamW).  fp32 mode.  Both the after-backward all-reduce (the default) and the
overlapped one.  Under gloo gan_amd.dp moves device tensors through the host itself (dp._host_staged says why).  tests/dp_two_rank_worker.py is the rank program."""
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_two_ranks(worker: str, args, tag: str, tmp_path):
    """Start `worker` twice (ranks 0 and 1 of a gloo group on 127.0.0.1), wait for both with a deadline, stop the peer of a rank that
    failed, and return the JSON rank 0 wrote to its last argument."""
    out = tmp_path / f"{tag}.json"
    port = _free_port()
    logdir = ROOT / "gpurun_out" if (ROOT / "gpurun_out").is_dir() else tmp_path  # on the GPU box the ranks' stderr survives the call
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / worker), *args, str(out)], cwd=str(ROOT), env=env,
                                      stdout=subprocess.DEVNULL, stderr=open(logdir / f"{tag}_rank{rank}.err", "w+"), text=True))
    deadline = time.time() + 120
    try:
        while any(p.poll() is None for p in procs) and time.time() < deadline and not any(p.poll() not in (None, 0) for p in procs):
            time.sleep(0.5)  # a rank that failed leaves its peer inside a collective: stop waiting as soon as one has
    finally:
        for p in procs:  # the exact children, never a pattern
            if p.poll() is None:
                p.kill()
    for p in procs:
        p.wait()
    outs = [(None, (logdir / f"{tag}_rank{i}.err").read_text()) for i in range(2)]
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} exit {p.returncode}\n" + "\n".join(f"--- rank {i} stderr ---\n{o[1][-2500:]}" for i, o in enumerate(outs))
    return json.loads(out.read_text())


@pytest.mark.parametrize("mode", ["after", "overlap"])
def test_two_ranks_equal_one_process_on_the_global_batch(mode, tmp_path):
    r = _run_two_ranks("dp_two_rank_worker.py", [mode], f"dp2_{mode}", tmp_path)
    msg = json.dumps(r)
    assert r["same_across_ranks"], msg                       # bit for bit: same gradients, same clip coefficient (dp.sync_sqnorm), same AdamW
    assert r["grad_d_vs_hand_averaged_shards"] <= 2e-6, msg  # the plumbing: the all-reduced D gradient IS the mean of the shards' gradients
    # against the whole batch in ONE process there is more: the decoder's fp32 summation order depends on the batch size (3e-7 on the
    # waveform), and LeakyReLU kinks in D turn that into 7e-5 on D's gradients, with or without data parallelism (the hand-averaged
    # shards show the same 7e-5; D alone on bit-identical inputs gives 6e-8 for 2 + 2 items against 4: tools/batch_split_check.py)
    assert r["grad_d_step1"] <= 1e-3 and r["grad_g_step1"] <= 3e-3, msg
    assert r["loss_g_mean_of_ranks"] == pytest.approx(r["loss_g_single"], rel=1e-5) and r["loss_d_mean_of_ranks"] == pytest.approx(r["loss_d_single"], rel=1e-5), msg
    # two AdamW steps later: a sanity bound only (first steps ~ lr * sign(g) turn 1e-4 differences in small gradients into lr-sized ones)
    assert r["param_g_step2"] <= 2e-2 and r["param_d_step2"] <= 2e-2 and r["param_g_l2"] <= 5e-2 and r["param_d_l2"] <= 1e-2, msg


def test_mel_stage1_two_ranks_equal_the_hand_averaged_shards(tmp_path):
    """The mel track's stage-1 iteration under data parallelism (SURVEY.md 8f-4 x 8e).  Its discriminator has BatchNorm, per rank as
    under DistributedDataParallel without SyncBN conversion, so the gradients are compared with the two shards run WITHOUT data
    parallelism and averaged by hand (D: the D step of iteration 0; G: an iteration without a D step and without the MR-STFT ratio);
    the MR-STFT term does not pass through BatchNorm and has to be the whole batch's value, not the shard's."""
    r = _run_two_ranks("mel_dp_two_rank_worker.py", [], "dp2_mel", tmp_path)
    msg = json.dumps(r)
    assert r["same_across_ranks"], msg
    assert r["grad_d_vs_hand_averaged_shards"] <= 2e-6 and r["grad_g_vs_hand_averaged_shards"] <= 2e-6, msg
    assert r["mrstft_dp"] == pytest.approx(r["mrstft_whole_batch"], rel=1e-5), msg
    assert abs(r["mrstft_rank0_shard_alone"] - r["mrstft_whole_batch"]) > 1e-4 * abs(r["mrstft_whole_batch"]), msg  # (the check can tell them apart)
    assert r["l1_dp_rank0"] == pytest.approx(r["l1_rank0_shard_alone"], rel=1e-6), msg                            # per-rank means stay per rank
