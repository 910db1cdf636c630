"""Run the reference's own, unmodified sources in this container.  TEST INFRASTRUCTURE ONLY.

`/root/reference` needs torch_geometric==1.0.3, torch_scatter==1.1.2, rdkit and tensorboardX, none of
which are installed (no network).  `install()` puts the stand-ins of `pyg103.py` into `sys.modules`;
`load("chem" | "bio")` then imports the reference's files from where they lie (nothing is copied) and
returns them as a namespace:

    ref = refshim.load("chem")
    ref.model.GNN, ref.batch.BatchMasking, ref.util.MaskAtom, ref.pretrain_masking.train, ...

The reference imports its siblings flat (`from model import GNN`, cwd = chem/ or bio/), and chem/ and bio/
use the same file names, so the flat names are bound in `sys.modules` only while one domain is being
imported and the modules are kept under `pgnn_ref_<domain>.<name>` afterwards.

Used by `oracle/refshim/make_fixtures.py` (writes tests/golden/ref_*) and by the live cross-checks in
tests/test_cpu_reference.py; both are skipped where /root/reference does not exist (the GPU box).
"""
import importlib.util
import os
import sys
import types

from . import pyg103

REFERENCE_ROOT = os.environ.get("PGNN_REFERENCE_ROOT", "/root/reference")
_FLAT = ("loader", "splitters", "batch", "dataloader", "util", "model", "pretrain_masking", "pretrain_contextpred",
         "pretrain_edgepred", "pretrain_deepgraphinfomax", "pretrain_supervised", "finetune")
_loaded = {}


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "chem", "model.py"))


def install():
    for name, mod in pyg103.build_modules().items():
        sys.modules.setdefault(name, mod)


def load(domain):
    """import /root/reference/<domain>/*.py (unmodified) and return them as attributes of a namespace"""
    if domain in _loaded:
        return _loaded[domain]
    if not available():
        raise FileNotFoundError("reference sources not found under %s" % REFERENCE_ROOT)
    install()
    root = os.path.join(REFERENCE_ROOT, domain)
    saved = {n: sys.modules.pop(n) for n in _FLAT if n in sys.modules}
    ns = types.SimpleNamespace(domain=domain, root=root)
    sys.path.insert(0, root)
    try:
        for name in _FLAT:
            path = os.path.join(root, name + ".py")
            if not os.path.isfile(path):
                continue
            if name in sys.modules:  # pulled in by an earlier sibling's flat import
                mod = sys.modules[name]
            else:
                spec = importlib.util.spec_from_file_location(name, path)
                mod = importlib.util.module_from_spec(spec)
                sys.modules[name] = mod
                spec.loader.exec_module(mod)
            setattr(ns, name, mod)
    finally:
        sys.path.remove(root)
        for name in _FLAT:
            mod = sys.modules.pop(name, None)
            if mod is not None:
                sys.modules["pgnn_ref_%s.%s" % (domain, name)] = mod
        sys.modules.update(saved)
    _loaded[domain] = ns
    return ns
