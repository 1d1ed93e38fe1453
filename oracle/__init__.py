"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (plain PyTorch fp32/fp64 functional code, no nn.Module, NCHW) of the
hot path of tqch/v-diffusion-torch:

  * ``unet_ref``       -- UNet.forward               (reference v_diffusion/models/unet.py:286-322)
  * ``diffusion_ref``  -- schedule / posterior / train_loss / p_sample
                          (reference v_diffusion/diffusion.py:42-545)
  * ``detrand``        -- build-owned deterministic tensor generator (integer hash, no torch RNG)
  * ``make_goldens``   -- imports the real reference from /root/reference (build container only)
                          and writes tests/golden/*.npz

Parity status: PINNED.  ``make_goldens.py`` asserts this restatement equal to the imported
reference (same weights, same inputs) and commits the reference's own outputs as fixtures;
``tests/test_oracle_vs_golden.py`` re-checks the restatement against those fixtures on every run.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package -- as the checker / reported baseline, never as the product path.  The
product (``v-diffusion-torch_amd/v_diffusion``) never imports it and raises when the HIP
library is missing.
"""
