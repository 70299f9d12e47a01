"""CPU oracle for the DDPM hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a plain numpy restatement of the algorithm in the
reference's ``src/models/diffusion_model.py`` (each function cites the lines it
follows).  It exists to check the HIP path, never to serve it: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it.  The product package never imports this module and raises when its HIP
library is missing instead of falling back to anything here.

Parity pin: the reference's own tests hold no numbers for this path (they patch
the model with ``Mock``), so the oracle is pinned by golden vectors captured from
the reference itself (``tests/golden/make_golden.py`` imports
``/root/reference/src/models/diffusion_model.py`` and records its outputs for
weights and noise that this oracle can regenerate without the reference).
"""
