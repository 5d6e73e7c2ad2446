"""
``BoundPlate`` (a Plate + plate sizes + inputs + parameters) and ``Problem`` (P, Q, data) --
BoundPlate.py / Problem.py of the reference, restricted to what the ELBO path needs.  Parameters
declared with OptParam become ``nn.Parameter``s; QEMParam values are kept as buffers (sampling and
log-probs work) but the QEM optimiser itself is out of scope of this build.
"""
import torch as t
import torch.nn as nn

from .dims import PT, ExpPT, Dim, dims_of, named_to_dim, dim_to_named
from .model import Plate, check_name, flatten_tree, tensordict2tree
from .samplers import PermutationSampler, Sampler, on_device


class _NamedStore(nn.Module):
    """Dict of named tensors held as buffers or parameters (so .to(device) moves them)."""

    def __init__(self, tensors, as_parameters=False):
        super().__init__()
        self._keys, self._names = [], {}
        for k, v in tensors.items():
            assert isinstance(v, t.Tensor), f"{k} must be a tensor"
            assert not hasattr(self, f"t_{k}")
            self._keys.append(k)
            self._names[k] = v.names
            raw = v.rename(None)
            if as_parameters:
                self.register_parameter(f"t_{k}", nn.Parameter(raw.clone()))
            else:
                self.register_buffer(f"t_{k}", raw.clone())

    def to_dict(self):
        return {k: getattr(self, f"t_{k}").refine_names(*self._names[k]) for k in self._keys}


def _plate_names(x):
    return [n for n in x.names if n is not None]


def named_to_pt(names, raw, platedims):
    """(names, unnamed tensor) -> PT with the plate dims leading (a view; permutes only if needed)."""
    lead = [i for i, n in enumerate(names) if n is not None]
    rest = [i for i, n in enumerate(names) if n is None]
    if lead + rest != list(range(len(names))):
        raw = raw.permute(*lead, *rest)
    return PT(raw, [platedims[names[i]] for i in lead])


def pt_tree(plate, flat, platenames_of):
    """Place each PT at the deepest plate it carries (cf. model.tensordict2tree, without torchdim)."""
    from .model import empty_tree
    root = empty_tree(plate)
    for name, p in flat.items():
        mine = {platenames_of[i] for i in p.ids if i in platenames_of}
        branch = root
        while mine:
            nxt = [q for q in mine if isinstance(branch.get(q), dict)]
            assert len(nxt) == 1, f"{name}: plates {mine} do not form a nested path"
            branch = branch[nxt[0]]
            mine.remove(nxt[0])
        branch[name] = p
    return root


def expand_named(x, names, all_platesizes):
    """Give a parameter's initial value every plate dim of the variable it belongs to."""
    have = _plate_names(x)
    for n in (*have, *names):
        if n not in all_platesizes:
            raise Exception(f"{n} is a plate dimension, but is not given in all_platesizes")
    extra = [n for n in names if n not in have]
    if not extra:
        return x.align_to(*names, ...) if have else x
    shape = [all_platesizes[n] for n in extra]
    out = x.rename(None).expand(*shape, *x.shape).contiguous()
    return out.refine_names(*extra, *x.names).align_to(*names, ...)


class BoundPlate(nn.Module):
    def __init__(self, plate, all_platesizes=None, inputs=None, extra_opt_params=None):
        super().__init__()
        self.register_buffer("_device_tensor", t.zeros(()))
        assert isinstance(plate, Plate)
        self.plate = plate
        all_platesizes = {} if all_platesizes is None else dict(all_platesizes)
        for pn in plate.all_platenames():
            if pn not in all_platesizes:
                raise Exception(f"Every plate must have a platesize specified in all_platesizes, but {pn} "
                                "doesn't have a specified size")
        self.all_platesizes = all_platesizes
        inputs = {} if inputs is None else dict(inputs)
        extra_opt_params = {} if extra_opt_params is None else dict(extra_opt_params)
        for k, v in {**inputs, **extra_opt_params}.items():
            if not isinstance(v, t.Tensor):
                raise Exception(f"`inputs` and `extra_opt_params` must be provided as a plain named tensor, "
                                f"but {k} is of type {type(v)}")
            for n in _plate_names(v):
                if n not in all_platesizes:
                    raise Exception(f"Dimension name {n} used on input/extra_opt_param {k}, but not provided "
                                    "in all_platesizes")
                if v.size(n) != all_platesizes[n]:
                    raise Exception(f"Dimension mismatch for input {k} along dimension {n}; all_platesizes "
                                    f"gives {all_platesizes[n]}, while {k} is {v.size(n)}")

        g2p = plate.groupvarname2platenames()
        opt, qem = dict(extra_opt_params), {}
        self.opt_paramname2trans = {k: (lambda x: x) for k in opt}
        for varname, (groupname, dist) in plate.varname2groupvarname_dist().items():
            for pname, (argname, param) in dist.opt_qem_params.items():
                if pname in opt or pname in qem:
                    raise Exception(f"Param is trying to add parameter named {pname}, but there's already a "
                                    "parameter with this name")
                value = expand_named(param.init, g2p[groupname], all_platesizes)
                if dist.qem_dist:
                    qem[pname] = value
                else:
                    opt[pname] = value
                    self.opt_paramname2trans[pname] = param.trans

        self._inputs = _NamedStore(inputs)
        self._opt_params = _NamedStore(opt, as_parameters=True)
        self._qem_params = _NamedStore(qem)
        self._dists = nn.ModuleDict({k: d for k, d in plate.varname2dist().items()})

        names = [*inputs, *opt, *qem]
        for n in names:
            check_name(n)
        if len(set(names)) != len(names):
            raise Exception("BoundPlate has overlapping names in inputs, opt_params, and/or qem_params")
        clash = set(names).intersection(plate.all_prog_names())
        if clash:
            raise Exception("The program in BoundPlate has plate/random variable names that overlap with the "
                            f"inputs/params.  Specifically {clash}.")
        self.sample()      # checks all dependencies resolve

    @property
    def device(self):
        return self._device_tensor.device

    def inputs(self):
        return self._inputs.to_dict()

    def opt_params(self):
        return {k: self.opt_paramname2trans[k](v) for k, v in self._opt_params.to_dict().items()}

    def qem_params(self):
        return self._qem_params.to_dict()

    def inputs_params_flat_named(self):
        return {**self.inputs(), **self.opt_params(), **self.qem_params()}

    def inputs_params(self, all_platedims):
        flat = {k: named_to_dim(v, all_platedims) for k, v in self.inputs_params_flat_named().items()}
        return tensordict2tree(self.plate, flat)

    def inputs_params_flat_pt(self, all_platedims):
        """Inputs and (transformed) parameters as PTs -- no named-tensor or torchdim ops on this path."""
        out = {}
        for store, trans in ((self._inputs, None), (self._opt_params, self.opt_paramname2trans),
                             (self._qem_params, None)):
            for k in store._keys:
                raw = getattr(store, f"t_{k}")
                if trans is not None and trans[k] is t.exp:
                    p = named_to_pt(store._names[k], raw, all_platedims)
                    out[k] = ExpPT(p.x, p.dims)            # exp() applied lazily (or inside the producer)
                    continue
                if trans is not None:
                    raw = trans[k](raw)
                out[k] = named_to_pt(store._names[k], raw, all_platedims)
        return out

    def groupvarname2platenames(self):
        return self.plate.groupvarname2platenames()

    def varname2groupvarname(self):
        return self.plate.varname2groupvarname()

    def _sample(self, K, reparam, sampler, all_platedims):
        assert isinstance(K, int) and isinstance(reparam, bool) and issubclass(sampler, Sampler)
        Kdims = self.plate.groupvarname2Kdim(K)
        platenames_of = {id(d): n for n, d in all_platedims.items()}
        ip = pt_tree(self.plate, self.inputs_params_flat_pt(all_platedims), platenames_of)
        from . import dist as D
        from .dims import PendingPT
        batch = D._DrawBatch() if D.BATCH_DRAWS else None
        saved, D._DRAW_BATCH[0] = D._DRAW_BATCH[0], batch
        try:
            with on_device(self.device):
                tree = self.plate.sample(None, {}, ip, [], all_platedims, Kdims, sampler, reparam)
        finally:
            D._DRAW_BATCH[0] = saved
        if batch is not None:
            batch.flush()

            def settle(tr):
                return {k: (settle(v) if isinstance(v, dict) else v.settled() if isinstance(v, PendingPT) else v)
                        for k, v in tr.items()}
            tree = settle(tree)
        return tree, Kdims

    def sample(self, sample_size=1):
        """One (or N) joint sample(s) from the model as a flat dict of named tensors."""
        all_platedims = {n: Dim(n, s) for n, s in self.all_platesizes.items()}
        platedims = list(all_platedims.values())
        tree, _ = self._sample(sample_size, False, PermutationSampler, all_platedims)
        out = {}
        N = Dim("N", sample_size)
        for k, v in flatten_tree(tree).items():
            v = v.dim()
            Ks = [d for d in dims_of(v) if d not in set(platedims)]
            v = v.order(*Ks) if Ks else v
            if sample_size > 1:
                v = v[N] if Ks else v
            else:
                for _ in Ks:
                    v = v.squeeze(0)
            out[k] = dim_to_named(v.detach(), order=[N, *platedims] if sample_size > 1 else platedims)
        return out


def _same_structure(name, P, Q, data_tree):
    """P and Q must list the same variables/plates; Q marks observed variables with Data() and
    ``data`` must provide exactly those (checking.py of the reference, condensed)."""
    from .model import Data
    if set(P.flat_prog) != set(Q.flat_prog):
        raise Exception(f"P and Q have different variables in plate {name}: "
                        f"{sorted(set(P.flat_prog) ^ set(Q.flat_prog))}")
    for k, q in Q.flat_prog.items():
        p = P.flat_prog[k]
        if isinstance(q, Plate):
            if not isinstance(p, Plate):
                raise Exception(f"{k} is a Plate in Q but not in P")
            _same_structure(k, p, q, data_tree.get(k, {}))
        elif isinstance(q, Data):
            if k not in data_tree:
                raise Exception(f"{k} is Data() in Q but no data was provided for it")
        elif k in data_tree:
            raise Exception(f"data provided for {k}, which Q treats as a latent variable")


class Problem(nn.Module):
    def __init__(self, P, Q, data):
        super().__init__()
        if not isinstance(P, BoundPlate) or not isinstance(Q, BoundPlate):
            raise Exception("P and Q must be `BoundPlate`s, not e.g. `Plate`s.  You can convert just using "
                            "`bound_plate_P = BoundPlate(plate_P)` if it doesn't have any inputs or parameters")
        self.register_buffer("_device_tensor", t.zeros(()))
        self.P, self.Q = P, Q
        if P.all_platesizes != Q.all_platesizes:
            raise Exception(f"all_platesizes does not match between P and Q.  In P it is {P.all_platesizes}, "
                            f"while in Q it is {Q.all_platesizes}")
        self.all_platedims = {n: Dim(n, s) for n, s in P.all_platesizes.items()}
        self._data = _NamedStore(data)
        _same_structure(None, P.plate, Q.plate, self.data)
        pin, qin = P.inputs_params_flat_named(), Q.inputs_params_flat_named()
        for k in set(pin) & set(qin):
            if pin[k].shape != qin[k].shape or not t.equal(pin[k].rename(None), qin[k].rename(None)):
                raise Exception(f"input/parameter {k} is defined differently on P and Q")

    @property
    def data(self):
        flat = {k: named_to_dim(v, self.all_platedims) for k, v in self._data.to_dict().items()}
        return tensordict2tree(self.P.plate, flat)

    @property
    def device(self):
        return self._device_tensor.device

    # ---- what a captured evaluation reads (Sample.elbo_nograd's automatic HIP-graph replay) -------------------------
    def memory_fingerprint(self):
        """Addresses (and sizes) of every parameter and buffer of the problem, walked afresh on every call: a captured
        evaluation stays valid for as long as these do not change (in-place updates -- optimiser steps, load_state_dict
        -- keep them; a sub-module's .to() / .double(), load_state_dict(assign=True) or a replaced Parameter do not, and
        a cached tensor list would have kept the old tensors alive and their addresses unchanged)."""
        # (the tensors are read from every module's own _parameters / _buffers dicts on every call; what is remembered is
        # only the LIST of modules -- nn.Module.parameters() spends most of its time re-walking the module tree, a third of
        # a replayed evaluation's host time -- and that list is rebuilt whenever a module gained or lost a child)
        # -- or had one REPLACED by another module: the children are compared by identity, not counted)
        mods = self.__dict__.get("_fp_modules")
        if mods is None or tuple(id(c) for m in mods for c in m._modules.values()) != self.__dict__.get("_fp_children"):
            mods = list(self.modules())
            self.__dict__["_fp_modules"] = mods
            self.__dict__["_fp_children"] = tuple(id(c) for m in mods for c in m._modules.values())
        out, seen = [], set()
        for m in mods:
            for x in (*m._parameters.values(), *m._buffers.values()):
                if x is not None and id(x) not in seen:
                    seen.add(id(x))
                    out.append((x.data_ptr(), x.numel()))
        return tuple(out)

    def check_device(self):
        if not (self.device == self.P.device and self.device == self.Q.device):
            raise Exception("Device issue: Problem, P and/or Q aren't all on the same device.  The easiest way "
                            "to make sure everything works is to call e.g. problem.to('cuda'), rather than "
                            "e.g. P.to('cuda').")

    def inputs_params(self):
        flat = {**self.P.inputs_params_flat_named(), **self.Q.inputs_params_flat_named()}
        flat = {k: named_to_dim(v, self.all_platedims) for k, v in flat.items()}
        return tensordict2tree(self.P.plate, flat)

    # ---- PT views used by the ELBO path (Sample._elbo) -------------------------------------
    def _platenames_of(self):
        return {id(d): n for n, d in self.all_platedims.items()}

    def inputs_params_pt(self):
        flat = {**self.P.inputs_params_flat_pt(self.all_platedims),
                **self.Q.inputs_params_flat_pt(self.all_platedims)}
        return pt_tree(self.P.plate, flat, self._platenames_of())

    def data_pt(self):
        flat = {k: named_to_pt(self._data._names[k], getattr(self._data, f"t_{k}"), self.all_platedims)
                for k in self._data._keys}
        return pt_tree(self.P.plate, flat, self._platenames_of())

    def sample(self, K, reparam=True, sampler=PermutationSampler):
        """K samples of every latent from Q, each latent group on its own K dim (Problem.py:71-97)."""
        from .sample import Sample
        self.check_device()
        tree, Kdims = self.Q._sample(K, reparam, sampler, self.all_platedims)
        return Sample(problem=self, sample=tree, groupvarname2Kdim=Kdims, sampler=sampler, reparam=reparam)
