"""CPU oracle for the GIN/GCN message-passing hot path of snap-stanford/pretrain-gnns.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pretrain_gnns_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

The oracle is a pure-PyTorch (CPU, fp32) restatement of the reference's
``chem/model.py`` and ``bio/model.py`` hot path plus the ``train()`` bodies of the
masking / context-prediction pre-training scripts.  The reference itself cannot be
imported in this image: it needs ``torch_geometric==1.0.3`` and
``torch_scatter==1.1.2`` (``requirements.txt:4-5``), which are absent, so the three
PyG-1.0.3 primitives it uses are restated in ``oracle/pyg_semantics.py``.

PARITY PINNING.  The reference has no test that pins ``GNN.forward`` numerically
(SURVEY.md §4, §8c): **parity unpinned** at the arithmetic level.  What *is* pinned:
  * the state-dict key/shape contract and real GCN / GraphSAGE / GAT weights + BN running statistics
    through the shipped ``chem|bio/model_architecture/{gcn,graphsage,gat}_*.pth`` checkpoints, which
    the oracle strict-loads (``oracle/make_golden.py`` -> ``tests/golden/*.pt``; the GIN blobs are
    absent from the reference repository);
  * the edge-ordering / masking contracts of ``chem/util.py:212-213,229-241``.
"""
