"""The fused level kernels of csrc/chain.hip are chosen by level size (one to two rounds of the chip:
BASELINE cfg 2 / cfg 4), so the small parity cases of test_gpu_parity.py run the GEMM + pointwise
launches instead.  Here the same parity cases (reference golden vectors, oracle on random / ragged /
64-vertex batches) are re-run in a child process with the fused kernels FORCED for every level size,
in each of their modes: 8-wave workgroups, 4-wave workgroups, both encoder directions in one grid, the
encoder's forward chain as one persistent launch, the 32-row kernels of small levels and the 16-row kernels of the
smallest (each forced for every size; the 16-row kernels once as per-level launches and once as the persistent
small-batch chains of csrc/chain_persist.hip, which are the default up to 128 rows), and the loss forward's per-level side-stream products and the decoder's per-level
weight-gradient terms, which the step uses above 256 / 64 graphs only (once more with the ENCODER's per-level terms on
the third, "aux" stream: off by default, measured slower, kept as a knob).
(The knobs are read once per process, hence the child process; one at a time, GPU use stays at 2 processes.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORCED = {"DVAE_CHAIN_FUSED": "2", "DVAE_CHAIN_FUSED_BWD": "2"}
CASES = ("tiny_matches_reference_golden or random_graphs_match_oracle or max_size_graphs_n64 or edge_cases_smallest "
         "or (hs501_matches_reference_golden and cfg4)")


# the test batches are small: keep them off chain_small_kernel / chain_tiny_kernel for the 64-row kernels' modes
BIG = {"DVAE_CHAIN_SMALL_M": "0", "DVAE_CHAIN_TINY_M": "0"}


@pytest.mark.parametrize("extra", [BIG, dict(BIG, DVAE_CHAIN_KS="2"), dict(BIG, DVAE_CHAIN_MERGE="1"),
                                   dict(BIG, DVAE_CHAIN_PERSIST="1"), {"DVAE_CHAIN_SMALL_M": "4096", "DVAE_CHAIN_TINY_M": "0"},
                                   {"DVAE_CHAIN_SMALL_M": "4096", "DVAE_CHAIN_TINY_M": "4096", "DVAE_CHAIN_TINY_PERSIST": "0"},
                                   {"DVAE_CHAIN_SMALL_M": "4096", "DVAE_CHAIN_TINY_M": "4096", "DVAE_CHAIN_TINY_PERSIST": "1"},
                                   {"DVAE_BY_LEVEL": "2", "DVAE_WGRAD_BY_LEVEL": "9", "DVAE_CHAIN_TINY_PERSIST": "0"},
                                   {"DVAE_BY_LEVEL": "2", "DVAE_WGRAD_BY_LEVEL": "11", "DVAE_CHAIN_TINY_PERSIST": "0"}],
                         ids=["8-wave", "4-wave", "merged-directions", "persistent-encoder", "small-level-kernels",
                              "tiny-level-kernels", "tiny-persistent-chains", "per-level-side-stream",
                              "per-level-aux-stream"])
def test_parity_with_fused_level_kernels_forced(extra):
    env = dict(os.environ, **FORCED, **extra)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", CASES, "-p", "no:cacheprovider", "--timeout", "120", "--timeout-method", "thread"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("extra", [{"DVAE_GATES_BF16": "0"}, {"DVAE_GATES_BF16": "1", "DVAE_GRADS_BF16": "0"}, {}],
                         ids=["fp32-gates", "half-gates", "half-gates-and-grads"])
def test_bf16_parity_with_activation_shadows_forced(extra):
    """bf16 mode above 256 graphs per batch, levels too large for the fused level kernels (cfg 5): the chains run as
    product + pointwise launches, states / predecessor sums / d gh / d gm get bf16 shadows that the products read
    (csrc/train.hip: act_shadows), and -- DVAE_GATES_BF16, default on -- the pre-activations and the saved gates are
    stored as bf16 (half_gates) and d gh / d gm exist as bf16 only (half_grads, DVAE_GRADS_BF16, default on: bf16-stored
    K-major operands on every weight-gradient tile, bf16 column sums).  The parity batches are smaller: DVAE_ACT_BF16=2 forces the shadows at every size and
    DVAE_CHAIN_FUSED*=0 keeps the small levels off the fused kernels (which would bypass both), and the bf16-mode
    parity cases must hold."""
    env = dict(os.environ, DVAE_ACT_BF16="2", DVAE_CHAIN_FUSED="0", DVAE_CHAIN_FUSED_BWD="0", **extra)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", "cfg5_shape_bf16_mode_close_to_oracle or bf16_mode_close_to_oracle", "-p", "no:cacheprovider",
                        "--timeout", "120", "--timeout-method", "thread"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("rows", ["64", "128"])
def test_bf16_parity_with_bf16_level_kernels_forced(rows):
    """(rows: the kernels' row tile, DVAE_CHAIN_BF16_ROWS: 64 = two workgroups per CU, the launch default; 128 = round 2's.)
    The bf16 level kernels (chain_bf16_kernel; off by default inside the train step) forced for every level size
    (DVAE_CHAIN_BF16=2): the bf16-mode parity cases must hold with them -- the orchestration then finds the bf16 weight
    copies (straight for the forward operations, transposed for the backward ones) through the call's context."""
    env = dict(os.environ, DVAE_CHAIN_BF16="2", DVAE_CHAIN_BF16_ROWS=rows)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", "test_gpu_parity.py"), os.path.join("tests", "test_gpu_chain_bf16.py"),
                        "-q", "-x", "-m", "gpu",
                        "-k", "cfg5_shape_bf16_mode_close_to_oracle or bf16_mode_close_to_oracle or chain_bf16", "-p", "no:cacheprovider",
                        "--timeout", "120", "--timeout-method", "thread"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
