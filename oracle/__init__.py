"""TEST INFRASTRUCTURE ONLY — CPU oracle for the SegmenTron dense-convolution hot path.

Nothing under ``oracle/`` is part of the shipped product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only
as the checker.  The product (``segmentron_amd``) raises if its HIP library is missing; it
never routes through this package.

Contents
--------
* ``ref_import``  – imports the *reference* (``/root/reference``) with in-memory stubs for the
  uninstalled ``torchvision``/``thop`` packages.  Only usable in the dev container.
* ``synth``       – deterministic synthetic weights / inputs keyed by ``state_dict`` name, so the
  reference, the oracle and the HIP model can all be loaded with identical parameters without
  shipping a 164 MB checkpoint.
* ``torch_ref``   – plain ``torch.nn.functional`` CPU fp32 restatement of the reference's module
  graph for the hot path (DeepLabv3+ / Xception65), each function citing the reference
  file:line it follows.  Pinned against the reference itself by ``gen_golden.py`` →
  ``tests/golden/*.npz`` (the reference ships no tests or golden vectors of its own —
  SURVEY.md §4 — so "outputs of the reference run here" is the pin).
* ``gen_golden``  – the script that produced ``tests/golden`` (committed with the fixtures).
"""
