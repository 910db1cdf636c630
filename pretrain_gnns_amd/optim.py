"""``Adam``: torch.optim.Adam's update for the optimizers of the reference's train loops, as ONE kernel launch.

The reference builds one ``optim.Adam`` per module (chem/pretrain_masking.py:134-136: model, linear_pred_atoms,
linear_pred_bonds) with the same hyper-parameters and calls ``zero_grad()`` / ``step()`` on each.  ``Adam`` here has the
same constructor and the same two methods; ``Adam.shared([...modules...], lr=..., weight_decay=...)`` returns one handle
per module that share a single kernel launch, so the reference's loop body runs unchanged::

    optimizer_model, optimizer_linear_pred_atoms, optimizer_linear_pred_bonds = Adam.shared(
        [model.parameters(), linear_pred_atoms.parameters(), linear_pred_bonds.parameters()], lr=args.lr, weight_decay=args.decay)

The update is torch's formula in fp32 (csrc/optim.hip); the step counter lives on the device, so a step can be captured
in a HIP graph.  Parameters without a gradient are skipped, like torch does; their moments stay untouched.

Limits, enforced rather than silent:
  * ONE step counter serves every parameter, where torch keeps ``state['step']`` per parameter and starts it at that
    parameter's first gradient.  A parameter that never gets a gradient (the bond head without --mask_edge) is simply skipped,
    which is torch's behaviour; a parameter whose FIRST gradient shows up after other parameters have already been stepped
    would get torch's bias correction for the wrong t, so ``launch()`` raises for it.
  * lr / betas / eps / weight_decay are read from ``param_groups[0]`` of the handles at every launch (an LR scheduler or a manual
    ``param_groups[0]['lr'] = x`` takes effect); the handles of one ``Adam.shared`` share a launch and must agree on them.
  * every handle of ``Adam.shared`` must call ``step()`` in every round: stepping a handle twice, or calling ``zero_grad()``
    while a round is incomplete, raises (the update of ALL handles would silently never be issued otherwise).
"""
import ctypes

import torch

from . import _lib
from .ops import check, load, stream_ptr


class _Core:
    """the tensors of all handles, the flat moment buffers and the device step counter"""

    def __init__(self, params, lr, betas, eps, weight_decay):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.PgnnError("pretrain_gnns_amd.optim.Adam: contiguous fp32 GPU parameters only")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        self.offsets = offs
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_words = torch.zeros(32, dtype=torch.int64, device=dev)  # updates applied, the kernel's cache of beta^step, its arrival tickets
        self.step_count = self.step_words[0]
        self.max_tensors = int(load().pgnn_adam_max_tensors())
        self.waiting = 0  # handles that still have to call step() before the shared launch goes out
        self._tables, self._table_key = [], None
        self.handles = 0
        self.owners = []       # the Adam handles (their param_groups carry the hyper-parameters)
        self.stepped = set()   # ids of the handles that have stepped in the current round
        self.launches = 0
        self._seen = set()     # ids of the parameters that took part in the first launch

    def hyper(self):
        """(lr, beta1, beta2, eps, weight_decay) as the handles' param_groups say NOW"""
        vals = None
        for h in self.owners:
            g = h.param_groups[0]
            v = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))
            if vals is None:
                vals = v
            elif v != vals:
                raise _lib.PgnnError("pretrain_gnns_amd.optim.Adam.shared: the handles share ONE launch and must agree on lr / betas / eps / "
                                     "weight_decay (%r vs %r); use separate Adam instances for different hyper-parameters" % (vals, v))
        return vals if vals is not None else (self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay)

    @torch.no_grad()
    def launch(self):
        live = [(p, o) for p, o in zip(self.params, self.offsets) if p.grad is not None]
        ids = {id(p) for p, _ in live}
        if self.launches == 0:
            self._seen = ids
        elif not ids <= self._seen:
            raise _lib.PgnnError("pretrain_gnns_amd.optim.Adam: a parameter received its first gradient after %d updates of the others; the "
                                 "single device-side step count cannot give it torch's per-parameter bias correction" % self.launches)
        self.launches += 1
        lr, b1, b2, eps, wd = self.hyper()
        # the job tables (ctypes arrays of pointers / counts / offsets) are rebuilt only when a pointer moved: with the caching
        # allocator the gradients come back at the same addresses step after step, and building the tables was ~0.1 ms a step
        key = tuple([p.data_ptr() for p, _ in live] + [p.grad.data_ptr() for p, _ in live])
        if key != self._table_key:
            tables = []
            for i in range(0, len(live), self.max_tensors):
                part = live[i:i + self.max_tensors]
                n = len(part)
                for p, _ in part:
                    if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.device != p.device:
                        raise _lib.PgnnError("pretrain_gnns_amd.optim.Adam: gradients must be contiguous fp32 on the parameter's device")
                tables.append(((ctypes.c_void_p * n)(*[p.data_ptr() for p, _ in part]),
                               (ctypes.c_void_p * n)(*[p.grad.data_ptr() for p, _ in part]),
                               (ctypes.c_int64 * n)(*[p.numel() for p, _ in part]), (ctypes.c_int64 * n)(*[o for _, o in part]), n))
            self._tables, self._table_key = tables, key
        lib, sp = load(), stream_ptr()
        for i, (P, G, C, O, n) in enumerate(self._tables):
            # every part reads the same step count: only the last part's launch advances it
            counter = self.step_words if i == len(self._tables) - 1 else self.step_words.clone()
            check(lib.pgnn_adam_step(P, G, C, O, n, self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), counter.data_ptr(), lr,
                                     b1, b2, eps, wd, sp), "pgnn_adam_step")


class Adam:
    """torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0): zero_grad() and step()."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, _core=None):
        self._own = list(params)
        self._core = _core if _core is not None else _Core(self._own, lr, betas, eps, weight_decay)
        self._core.handles += 1
        self._core.owners.append(self)
        self.param_groups = [{"params": self._own, "lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}]

    @classmethod
    def shared(cls, param_lists, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        """one handle per parameter list; the update of all of them is a single launch, issued by the LAST handle whose
        ``step()`` is called in a round (the reference calls them back to back)"""
        lists = [list(ps) for ps in param_lists]
        core = _Core([p for ps in lists for p in ps], lr, betas, eps, weight_decay)
        return [cls(ps, lr, betas, eps, weight_decay, _core=core) for ps in lists]

    def zero_grad(self, set_to_none=True):
        if self._core.waiting != 0:
            raise _lib.PgnnError("pretrain_gnns_amd.optim.Adam.shared: zero_grad() with %d of %d handles not stepped -- the shared update "
                                 "has NOT been issued; every handle must call step() each round" % (self._core.waiting, self._core.handles))
        for p in self._own:
            if p.grad is None:
                continue
            if set_to_none:
                p.grad = None
            else:
                if p.grad.grad_fn is not None:
                    p.grad.detach_()
                else:
                    p.grad.requires_grad_(False)  # (a gradient that is a view -- parallel.GradBucket's -- cannot be detached in place)
                p.grad.zero_()

    def step(self):
        core = self._core
        if core.waiting == 0:
            core.waiting = core.handles
            core.stepped.clear()
        if id(self) in core.stepped:
            raise _lib.PgnnError("pretrain_gnns_amd.optim.Adam.shared: this handle stepped twice before the others stepped once; every "
                                 "handle must call step() each round (the update is one launch for all of them)")
        core.stepped.add(id(self))
        core.waiting -= 1
        if core.waiting == 0:
            core.launch()

    @property
    def step_count(self):
        return self._core.step_count
