"""CPU oracle for the Rot-MVGaze hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement of the reference's
algorithm (torch-CPU functional ops + numpy) that the parity tests, the
``__graft_entry__.smoke()`` check and ``bench.py``'s ``cpu_baseline`` leg use as
the *checker*.  Nothing under ``rot-mvgaze_amd/`` (the product path) may import
it: the product computes with the HIP library only and fails loudly when that
library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference's
own Python (``/root/reference``) in the build container, runs it on the seeded
inputs produced by ``oracle/synth.py`` and commits the outputs as fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against those fixtures.
"""
