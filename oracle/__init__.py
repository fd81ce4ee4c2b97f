"""CPU oracle for the MSGM / sdeflow-light hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32 or
fp64) *restatement* of the reference algorithm on the path named by
BASELINE.json:north_star.  It exists so the HIP kernels can be checked
against something; it is never the thing that is shipped or measured.

Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``sdeflow_light_amd/``
imports it (``tests/test_boundary.py`` greps for that).

Parity pin: every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors under ``tests/golden/`` that were produced by
importing the actual reference (``/root/reference``) in the build container
with ``tools/make_golden.py`` (plotting-only dependencies stubbed; see
SURVEY.md App. C).  The reference itself holds no tests or fixtures
(SURVEY.md §4), so those generated vectors are the only pin.

Layout
------
sde_ref.py   schedules, drift/diffusion (SGM, MSGM dense/sparse), EM/Heun/RK4
             integrators, forward perturbation, timestep indexing
nets_ref.py  functional score nets driven by a reference-keyed state_dict
             (MLP, UNet1D, 2-D UNet) + their building blocks
ssm_ref.py   sliced-score-matching loss (double-backward form as upstream
             and forward-mode JVP form as the HIP path computes it), Adam
"""
