"""CPU oracle for the deepbedmap ESRGAN hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of what the reference's srgan_train.py does
through Chainer 7.0.0 (reference pins: Pipfile:8 chainer==7.0.0, Pipfile:10
cupy-cuda100==7.0.0, Pipfile:21 numpy==1.17.3, Pipfile:30 ssim-chainer@9c54f25).
Chainer / CuPy / ssim-chainer are third-party packages that are NOT vendored under
/root/reference and are not importable in this image, so the arithmetic is restated
from their published algorithms and anchored on the reference's
own call sites, doctests and known-answer values.

Pinning status
--------------
* PINNED by the reference's own known answers (tests/test_oracle_kats.py):
  calculate_discriminator_loss -> 1.56670504 (srgan_train.py:985-991),
  calculate_generator_loss -> 4.35108415 (srgan_train.py:859-868),
  psnr -> 192.65919722494797 (srgan_train.py:916-920),
  ssim_loss_func -> 0.800004 (srgan_train.py:944-948),
  GeneratorModel().count_params() == 8907749, output (1,1,36,36) (srgan_train.py:437-447),
  DiscriminatorModel().count_params() == 10370761, output (2,1) (srgan_train.py:601-608),
  Y.shape / (X.shape - 2) == 4 (features/steps/test_deepbedmap.py:35-39).
* PARITY UNPINNED: the numeric output of GeneratorModel.forward / DiscriminatorModel.forward,
  gradients, Adam and BatchNorm running statistics have no golden vector anywhere in the
  reference (random unseeded inputs, no trained .npz in the repo) and Chainer cannot be run
  here.  For those, this oracle is cross-checked against an independent torch-CPU autograd
  restatement and finite differences (tests/test_oracle_vs_torch.py), not against Chainer.
  The SSIM window (uniform vs Gaussian sigma=1.5) is also unpinned: both reproduce the two
  constant-image known answers; the default here is Gaussian sigma=1.5.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package -- always as the checker, never as the product path.  The product
(deepbedmap_amd) never imports it and fails loudly when its HIP library is missing.
"""
from . import ops, model, train  # noqa: F401  (torch_ref is imported on demand: it needs torch)
