"""TEST INFRASTRUCTURE ONLY — never imported by the shipped package.

This directory holds a CPU (PyTorch fp32) restatement of the reference
denoising path, the closed-form weight generator shared by the oracle and the
HIP path in parity tests, and the script that pins the restatement against the
real reference (``make_golden.py``).  Only ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py`` may import from here, and only as
the checker — never as the thing that is measured or shipped.
"""
