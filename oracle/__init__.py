"""CPU oracle for the Groth16 proving hot path (TEST INFRASTRUCTURE, NOT PRODUCT).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import or execute anything under ``oracle/``.  The shipped path (``zokrates_amd`` +
``libzkhip.so``) never imports it and fails loudly when the HIP library is missing.

Parity status: **G16 proof bytes are "parity unpinned" by the reference's own tests**
(SURVEY.md §0.7, §8c): the arithmetic lives in crates.io ``ark-groth16 0.3.0`` /
``ark-poly 0.3.0`` / ``ark-ec 0.3.0`` / ``ark-ff 0.3.0`` (pinned in
``/root/reference/Cargo.lock:79-378``; source not vendored, no Rust toolchain), so the
reference cannot be run here.  What pins this oracle instead:

* in-tree known answers (``tests/golden/``: field KATs of ``zokrates_field/src/bn128.rs``,
  curve constants of ``zokrates_proof_systems/src/solidity.rs``, BN254 point fixtures from
  ``zokrates_cli/examples/book/mpc_tutorial/phase1radix2m2``);
* three mutually independent restatements that must agree bit-for-bit:
  O1 closed-form "trapdoor" proof (pure Fr arithmetic + 3 scalar mults),
  O2 the algorithmic restatement of ``ark_groth16::create_random_proof``
  (sparse mat-vec, 7 radix-2 transforms, 5 MSMs),
  O3 the pairing check implementing the verification equation of
  ``zokrates_proof_systems/src/scheme/groth16.rs:156-172``;
* uniqueness of a Groth16 proof for fixed (pk, z, r, s).

GM17 (``gm17.py``, ``c/gm17.hpp``) has the same layers plus one reference artefact: the in-tree golden
(proof, verification key, inputs) triple ``zokrates_stdlib/tests/tests/snark/gm17.json`` (BLS12-377) verifies under the
restated equations.  ``ir.py`` restates the ``out`` program format and ark's variable allocation order (no in-tree
fixture exists for either: it is an independent reading of the same source files as the C++ reader).
"""
