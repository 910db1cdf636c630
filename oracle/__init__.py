"""CPU oracle for the GIN/GCN message-passing hot path of snap-stanford/pretrain-gnns.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pretrain_gnns_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

The oracle is a pure-PyTorch (CPU, fp32) restatement of the reference's ``chem/model.py`` and
``bio/model.py`` hot path plus the ``train()`` bodies of the masking / context-prediction pre-training
and fine-tuning scripts.  It exists because the reference cannot travel to the GPU box and because a
restatement is what ``bench.py``'s ``cpu_baseline`` leg can time there.

PARITY PINNING: **pinned to the reference's own code.**  ``oracle/refshim`` replaces the four missing
third-party imports (torch_geometric 1.0.3, torch_scatter 1.1.2 -- their published behaviour restated in
``refshim/pyg103.py`` --, rdkit and tensorboardX as inert placeholders) and then imports the UNMODIFIED
files under /root/reference.  ``python -m oracle.refshim.make_fixtures`` runs them -- ``model.GNN`` /
``GNN_graphpred`` forward + backward (fp32 and fp64), ``BatchMasking`` / ``BatchSubstructContext``
``.from_data_list``, ``MaskAtom`` / ``MaskEdge`` / ``ExtractSubstructureContextPair``, and the ``train()`` /
``eval()`` functions of pretrain_masking.py, pretrain_contextpred.py, finetune.py (chem and bio) -- and
writes tests/golden/ref_*.npz; ``python -m oracle.make_golden`` runs the reference model class on the
shipped checkpoints.  tests/test_cpu_reference.py holds the oracle to those fixtures (bit-exact forward,
bit-exact multi-step Adam trajectories at one CPU thread) and, in the build container, to the live
reference; tests/test_gpu_reference.py holds the HIP path to the same fixtures.  What stays restated
rather than executed: the third-party surface in ``refshim/pyg103.py`` (no wheel on disk, no network).
"""
