"""
Model specification: ``Plate`` / ``Group`` / ``Data`` (Plate.py:50-83, Group.py, Data.py of the
reference), plus sampling of a plate tree from Q.  A Plate is compiled once, at construction, into an
ordered list of *entries* -- ("group", name, {var: Dist}) | ("data", name) | ("plate", name, Plate)
-- which is what the sampling and log-prob recursions iterate over.
"""
from .dims import Dim, dims_of
from .dist import Dist, _DistSpec
from .timeseries import Timeseries

RESERVED_PREFIXES = ("K_",)


def check_name(name):
    if not isinstance(name, str) or not name.isidentifier():
        raise Exception(f"{name!r} is not a valid variable / plate name")


class Data:
    """Placeholder in Q for a variable that is observed (Data.py)."""


class Group:
    """Several variables sharing ONE K dimension (Group.py:44-53)."""

    def __init__(self, **kwargs):
        for varname, dist in kwargs.items():
            if not isinstance(dist, (_DistSpec, Timeseries)):
                raise Exception(f"{varname} in a Group should be a Dist or Timeseries, but is actually {type(dist)}")
        if len(kwargs) < 2:
            raise Exception("Groups only make sense if they have two or more random variables")
        self.prog = {k: v.finalize(k) for k, v in kwargs.items()}


class Plate:
    def __init__(self, **kwargs):
        self.grouped_prog = {}     # name -> {var: Dist|Data|Timeseries}  |  Plate
        self.flat_prog = {}        # var/plate name -> Dist|Data|Timeseries|Plate
        for name, v in kwargs.items():
            check_name(name)
            if isinstance(v, (_DistSpec, Timeseries)) and not isinstance(v, Dist):
                v = v.finalize(name)
            if isinstance(v, Plate):
                self.grouped_prog[name] = v
                self.flat_prog[name] = v
            elif isinstance(v, Group):
                self.grouped_prog[name] = dict(v.prog)
                self.flat_prog.update(v.prog)
            elif isinstance(v, (Dist, Data, Timeseries)):
                self.grouped_prog[name] = {name: v}
                self.flat_prog[name] = v
            else:
                raise Exception(f"{name} in a Plate must be a distribution, Group, Plate, Timeseries or Data(); "
                                f"got {type(v)}")
        names = self.all_prog_names()
        dup = sorted({n for n in names if names.count(n) > 1})
        if dup:
            raise Exception(f"Plate has duplicate names {dup}.")

    # ---- structure queries ---------------------------------------------------------------
    def entries(self):
        for name, v in self.grouped_prog.items():
            if isinstance(v, Plate):
                yield "plate", name, v
            elif any(isinstance(d, Data) for d in v.values()):
                assert len(v) == 1
                yield "data", name, v
            else:
                yield "group", name, v

    def grouped_get(self, d, name):
        v = self.grouped_prog[name]
        if isinstance(v, Plate):
            return d[name]
        return {k: d.get(k) for k in v}

    def all_prog_names(self):
        out = []
        for kind, name, v in self.entries():
            out.append(name)
            if kind == "plate":
                out += v.all_prog_names()
            elif len(v) >= 2:
                out += list(v)
        return out

    def all_platenames(self):
        out = []
        for kind, name, v in self.entries():
            if kind == "plate":
                out += [name, *v.all_platenames()]
        return out

    def groupvarname2Kdim(self, K):
        """One fresh K dim per latent group/variable, named K_<group> (Plate.py:217-230)."""
        out = {}
        for kind, name, v in self.entries():
            if kind == "group":
                out[name] = Dim(f"K_{name}", K)
            elif kind == "plate":
                out.update(v.groupvarname2Kdim(K))
        return out

    def varname2groupvarname_dist(self):
        out = {}
        for kind, name, v in self.entries():
            if kind == "group":
                for var, dist in v.items():
                    out[var] = (name, dist)
            elif kind == "plate":
                out.update(v.varname2groupvarname_dist())
        return out

    def varname2groupvarname(self):
        return {k: g for k, (g, _) in self.varname2groupvarname_dist().items()}

    def varname2dist(self):
        return {k: d for k, (_, d) in self.varname2groupvarname_dist().items()}

    def groupvarname2platenames(self, active=()):
        out = {}
        for kind, name, v in self.entries():
            if kind == "plate":
                out.update(v.groupvarname2platenames((*active, name)))
            else:
                out[name] = list(active)
        return out

    # ---- sampling from this plate (as Q) --------------------------------------------------
    def sample(self, name, scope, inputs_params, active_platedims, all_platedims, groupvarname2Kdim,
               sampler, reparam, dimcache=None):
        """Ancestral sampling of this plate (as Q): a tree of PTs (Plate.py:93-143)."""
        if name is not None:
            active_platedims = [*active_platedims, all_platedims[name]]
        scope = update_scope(scope, inputs_params)
        dimcache = {} if dimcache is None else dimcache
        out = {}
        for kind, child, v in self.entries():
            if kind == "group":
                drawn = sample_group(v, scope, active_platedims, groupvarname2Kdim[child], sampler, reparam,
                                     dimcache)
                out.update(drawn)
                scope.update(drawn)
            elif kind == "plate":
                out[child] = v.sample(child, scope, inputs_params.get(child, {}), active_platedims, all_platedims,
                                      groupvarname2Kdim, sampler, reparam, dimcache)
        return out


def update_scope(scope, tree):
    """New scope = old scope + the tensors (not sub-trees) of ``tree`` (Plate.py:292-303)."""
    scope = dict(scope)
    for k, v in tree.items():
        if not isinstance(v, dict):
            assert k not in scope, f"{k} is already in scope"
            scope[k] = v
    return scope


def sample_group(prog, scope, active_platedims, K_dim, sampler, reparam, dimcache=None):
    """Draw every variable of one group with the group's K dim (dist.py:23-72): parents are first
    re-indexed from their own K dims onto ``K_dim`` by the sampler (permutation / categorical).
    scope values and results are PTs."""
    from .dims import PT
    needed = {a for d in prog.values() for a in d.all_args} - set(prog) - {"prev"}
    for a in needed:
        if a not in scope:
            raise Exception(f"{a} is not in scope")
    local = sampler.resample_scope_pt({k: PT.of(v) for k, v in scope.items() if k in needed},
                                      active_platedims, K_dim)
    has_ts = any(isinstance(d, Timeseries) for d in prog.values())
    perm = sampler.perm(dims={K_dim, *active_platedims}, Kdim=K_dim) if has_ts else None
    out = {}
    for var, dist in prog.items():
        x = PT.of(dist.sample(local, reparam, active_platedims, K_dim, perm, dimcache))
        local[var] = x
        out[var] = x
    return out


# ---- flat dict <-> plate-shaped tree -------------------------------------------------------
def empty_tree(plate):
    return {n: empty_tree(v) for kind, n, v in plate.entries() if kind == "plate"}


def tensordict2tree(plate, flat):
    """Place each tensor at the deepest plate whose dims it carries (Plate.py:351-373)."""
    root = empty_tree(plate)
    platenames = set(plate.all_platenames())
    for name, x in flat.items():
        mine = platenames.intersection(str(d) for d in dims_of(x))
        branch = root
        while mine:
            nxt = [p for p in mine if isinstance(branch.get(p), dict)]
            assert len(nxt) == 1, f"{name}: plates {mine} do not form a nested path"
            branch = branch[nxt[0]]
            mine.remove(nxt[0])
        branch[name] = x
    return root


def flatten_tree(tree):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(flatten_tree(v))
        else:
            out[k] = v
    return out


def tree_tensors(tree):
    return {k: v for k, v in tree.items() if not isinstance(v, dict)}
