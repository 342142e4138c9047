"""Irreducible-representation bookkeeping for the MI355X Equiformer hot path.

Replaces the small part of `e3nn.o3.Irreps` the reference's model files use
(reference call sites: nets/graph_attention_transformer.py:765-779, nets/tensor_product_rescale.py:224-231).
Only even-parity (SE(3)) irreps are accepted by the kernels; odd parities parse (so reference strings such as
'16x1o' are understood) but are rejected where a layout is built.
"""
import re

_TOKEN = re.compile(r"^\s*(?:(\d+)\s*x\s*)?(\d+)([eo])\s*$")


class Irrep:
    __slots__ = ("l", "p")

    def __init__(self, l, p=1):
        if isinstance(l, Irrep):
            l, p = l.l, l.p
        elif isinstance(l, str):
            m = _TOKEN.match(l)
            if not m or m.group(1):
                raise ValueError("bad irrep %r" % (l,))
            l, p = int(m.group(2)), (1 if m.group(3) == "e" else -1)
        elif isinstance(l, tuple):
            l, p = l
        self.l, self.p = int(l), int(p)

    @property
    def dim(self):
        return 2 * self.l + 1

    def couple(self, other):
        """Degrees reachable from self (x) other with their parity."""
        return [Irrep(l, self.p * other.p) for l in range(abs(self.l - other.l), self.l + other.l + 1)]

    def __eq__(self, o):
        o = Irrep(o)
        return (self.l, self.p) == (o.l, o.p)

    def __hash__(self):
        return hash((self.l, self.p))

    def __repr__(self):
        return "%d%s" % (self.l, "e" if self.p == 1 else "o")


class Irreps:
    """Ordered list of (multiplicity, Irrep) pairs."""

    def __init__(self, spec=None):
        items = []
        if spec is None:
            pass
        elif isinstance(spec, Irreps):
            items = list(spec.items)
        elif isinstance(spec, str):
            for tok in filter(None, (t.strip() for t in spec.split("+"))):
                m = _TOKEN.match(tok)
                if not m:
                    raise ValueError("bad irreps token %r" % (tok,))
                items.append((int(m.group(1) or 1), Irrep(int(m.group(2)), 1 if m.group(3) == "e" else -1)))
        else:
            for mul, ir in spec:
                items.append((int(mul), Irrep(ir)))
        self.items = items

    # -- sequence protocol
    def __iter__(self):
        return iter(self.items)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def __eq__(self, o):
        o = Irreps(o)
        return [(m, ir.l, ir.p) for m, ir in self.items] == [(m, ir.l, ir.p) for m, ir in o.items]

    def __ne__(self, o):
        return not self.__eq__(o)

    def __hash__(self):
        return hash(tuple((m, ir.l, ir.p) for m, ir in self.items))

    def __add__(self, o):
        return Irreps(self.items + Irreps(o).items)

    def __mul__(self, n):
        return Irreps(self.items * int(n))

    __rmul__ = __mul__

    def __contains__(self, ir):
        ir = Irrep(ir)
        return any(ir == i for _, i in self.items)

    def __repr__(self):
        return "+".join("%dx%r" % (m, ir) for m, ir in self.items)

    # -- e3nn-like queries
    @property
    def dim(self):
        return sum(m * ir.dim for m, ir in self.items)

    @property
    def num_irreps(self):
        return sum(m for m, _ in self.items)

    @property
    def lmax(self):
        return max(ir.l for _, ir in self.items)

    def slices(self):
        out, i = [], 0
        for m, ir in self.items:
            out.append(slice(i, i + m * ir.dim))
            i += m * ir.dim
        return out

    def simplify(self):
        out = []
        for m, ir in self.items:
            if m == 0:
                continue
            if out and out[-1][1] == ir:
                out[-1] = (out[-1][0] + m, ir)
            else:
                out.append((m, ir))
        return Irreps(out)

    def sort_even_first(self):
        """Stable sort by (l, even before odd); returns (sorted irreps, p) with p[old index] = new index
        (reference: sort_irreps_even_first, nets/tensor_product_rescale.py:224-231)."""
        order = sorted(range(len(self.items)), key=lambda i: (self.items[i][1].l, -self.items[i][1].p, i))
        p = [0] * len(order)
        for new, old in enumerate(order):
            p[old] = new
        return Irreps([self.items[i] for i in order]), p

    @staticmethod
    def spherical_harmonics(lmax):
        return Irreps([(1, Irrep(l, (-1) ** l)) for l in range(lmax + 1)])

    def require_even(self):
        for _, ir in self.items:
            if ir.p != 1:
                raise NotImplementedError("only even-parity (SE(3)) irreps are supported by the HIP kernels: %r" % self)
        return self
