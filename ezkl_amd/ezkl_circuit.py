"""The ezkl gate set as data: what `GraphCircuit::configure_with_params` builds, restated over
`halo2_cs.ConstraintSystem` so that a prover here works on the constraint system of a real ezkl circuit instead of
hand-written gates.  Mirrors, with the reference's own names:

  ModelVars::new / VarTensor::{new_advice, constant_cols}        /root/reference/src/graph/vars.rs:444-486, src/tensor/var.rs:57-230
  BaseConfig::configure            (ADD/SUB/MULT, DOT.. gates)   src/circuit/ops/chip.rs:321-448, base.rs:9-133
  BaseConfig::configure_lookup     (static lookup tables)        chip.rs:452-615, table.rs:41-204
  BaseConfig::configure_range_check                              chip.rs:837-970, table.rs:323-395
  BaseConfig::configure_dynamic_lookup / configure_shuffles      chip.rs:619-714, 739-833
  Einsums::configure_universal     (contraction + RLC gates)     chip/einsum/mod.rs:60-94, 487-783
  GraphCircuit::configure_with_params (the order of all this)    src/graph/mod.rs:1945-2004, model.rs:1044-1121

Selector creation order, BTreeMap iteration order (BaseOp's derive(Ord) = declaration order) and the query order inside
every gate closure are kept, because halo2 derives the proof layout from them (halo2_cs.py).  Pinned on the reference's
fixture circuit (tests/assets/{settings.json, vk.key, pk.key, proof.json}) in tests/test_ezkl_circuit.py.
"""
from . import plonk as P
from .halo2_cs import ConstraintSystem
from .plonk import R

# BaseOp in declaration order (= Ord), src/circuit/ops/base.rs:10-20
DOT, DOTINIT, CUMPRODINIT, CUMPROD, ADD, MULT, SUB, SUMINIT, SUM = range(9)
OP_NAME = {DOT: "DOT", DOTINIT: "DOTINIT", CUMPRODINIT: "CUMPRODINIT", CUMPROD: "CUMPROD", ADD: "ADD", MULT: "MULT", SUB: "SUB",
           SUMINIT: "SUMINIT", SUM: "SUM"}
NUM_INPUTS = {DOT: 2, DOTINIT: 2, CUMPRODINIT: 1, CUMPROD: 1, ADD: 2, MULT: 2, SUB: 2, SUMINIT: 1, SUM: 1}
# (rotation offset, range) of the output query; the constraint is on the LAST queried output cell
OFFSET_RNG = {DOTINIT: (0, 1), DOT: (-1, 2), CUMPROD: (-1, 2), CUMPRODINIT: (0, 1), ADD: (0, 1), SUB: (0, 1), MULT: (0, 1), SUM: (-1, 2),
              SUMINIT: (0, 1)}
RESERVED_BLINDING_ROWS_PAD = 3                      # src/circuit/table.rs:29


def felt(i):
    """integer_rep_to_felt (src/fieldutils.rs:9-17)"""
    return i % R


def nonaccum_f(op, a, b):
    return a + b if op == ADD else a - b if op == SUB else a * b


def accum_f(op, prev, a, b):
    zero, one = P.const(0), P.const(1)
    def fold(items, init, f):
        acc = init
        for it in items:
            acc = f(acc, it)
        return acc
    if op == DOTINIT: return fold(zip(a, b), zero, lambda acc, ab: acc + ab[0] * ab[1])
    if op == DOT: return prev + fold(zip(a, b), zero, lambda acc, ab: acc + ab[0] * ab[1])
    if op == CUMPRODINIT: return fold(b, one, lambda acc, x: acc * x)
    if op == CUMPROD: return prev * fold(b, one, lambda acc, x: acc * x)
    if op == SUMINIT: return fold(b, zero, lambda acc, x: acc + x)
    if op == SUM: return prev + fold(b, zero, lambda acc, x: acc + x)
    raise ValueError(op)


class VarTensor:
    """blocks x inner columns of advice (src/tensor/var.rs:17-32)"""

    def __init__(self, inner, num_inner_cols, col_size):
        self.inner, self.num_inner_cols, self.col_size = inner, num_inner_cols, col_size

    @staticmethod
    def max_rows(cs, logrows):
        return (1 << logrows) - cs.blinding_factors() - 1

    @classmethod
    def new_advice(cls, cs, logrows, num_inner_cols, capacity, phase=0, blinded=True):
        max_rows = cls.max_rows(cs, logrows)
        max_assignments = max_rows * num_inner_cols
        modulo = capacity // max_assignments + 1
        modulo = (capacity + modulo) // max_assignments + 1     # room for the duplicated row at each column overflow
        inner = []
        for _ in range(modulo):
            blk = []
            for _ in range(num_inner_cols):
                col = cs.advice_column(phase, blinded)
                cs.enable_equality(col)
                blk.append(col)
            inner.append(blk)
        return cls(inner, num_inner_cols, max_rows)

    @classmethod
    def new_unblinded_advice(cls, cs, logrows, num_inner_cols, capacity):
        return cls.new_advice(cs, logrows, num_inner_cols, capacity, 0, False)

    @classmethod
    def new_advice_in_second_phase(cls, cs, logrows, num_inner_cols, capacity):
        return cls.new_advice(cs, logrows, num_inner_cols, capacity, 1, True)

    @staticmethod
    def constant_cols(cs, logrows, num_constants, module_requires_fixed=False):
        if num_constants == 0 and not module_requires_fixed:
            return []
        if num_constants == 0:
            modulo = 1
        else:
            max_rows = VarTensor.max_rows(cs, logrows)
            modulo = num_constants // max_rows + 1
            modulo = (num_constants + modulo) // max_rows + 1
        cols = []
        for _ in range(modulo):
            col = cs.fixed_column()
            cs.enable_constant(col)
            cols.append(col)
        return cols

    def num_blocks(self): return len(self.inner)
    def block_size(self): return self.col_size * self.num_inner_cols

    def cartesian_coord(self, linear):
        """(block, inner column, row) of a linear cell index (src/tensor/var.rs:319-327)"""
        x = linear // self.block_size()
        return x, linear % self.num_inner_cols, (linear - x * self.block_size()) // self.num_inner_cols

    def query_rng(self, meta, x, y, z, rng):
        return [meta.query_advice(self.inner[x][y], z + i) for i in range(rng)]

    def query_whole_block(self, meta, x, z, rng):
        return [meta.query_advice(self.inner[x][y], z + i) for i in range(rng) for y in range(self.num_inner_cols)]


class SelectorConstructor:
    """src/circuit/table.rs:41-89"""

    def __init__(self, degree):
        self.degree = degree

    def get_expr_at_idx(self, i, expr):
        acc = P.const(1)
        for x in range(self.degree):
            if x != i:
                acc = acc * (expr if x == 0 else (P.const(x) - expr))
        return acc

    def get_selector_val_at_idx(self, i):
        acc = 1
        for x in range(self.degree):
            if x != i:
                acc = acc * (i if x == 0 else (x - i)) % R
        return acc


class RangeCheck:
    """src/circuit/table.rs:323-395: the values lo..=hi laid out over ceil columns of col_size rows, each column scaled
    by its selector value"""

    def __init__(self, cs, rng, logrows):
        self.range = rng
        self.col_size = (1 << logrows) - (cs.blinding_factors() + RESERVED_BLINDING_ROWS_PAD)
        num_cols = abs(rng[1] - rng[0]) // self.col_size + 1
        self.inputs = [cs.lookup_table_column() for _ in range(num_cols)]
        self.selector_constructor = SelectorConstructor(num_cols)

    def get_first_element(self, chunk):
        return felt(chunk * self.col_size + self.range[0])

    def columns(self, n, usable):
        """fixed-column assignments (ints), what `layout` assigns + halo2's table padding with the first row"""
        vals = list(range(self.range[0], self.range[1] + 1))
        return _table_columns(n, usable, self.col_size, self.selector_constructor, [vals])[0]


class Table:
    """static lookup table of a nonlinearity f over [lo, hi] (src/circuit/table.rs:91-321)"""

    def __init__(self, cs, rng, logrows, f, shared_inputs):
        self.range, self.f = rng, f
        self.col_size = (1 << logrows) - (cs.blinding_factors() + RESERVED_BLINDING_ROWS_PAD)
        num_cols = abs(rng[1] - rng[0]) // self.col_size + 1
        while len(shared_inputs) < num_cols:
            shared_inputs.append(cs.lookup_table_column())
        self.table_inputs = list(shared_inputs)
        self.table_outputs = [cs.lookup_table_column() for _ in self.table_inputs]
        self.selector_constructor = SelectorConstructor(len(self.table_inputs))

    def largest(self):
        return self.range[0] + self.col_size * len(self.table_inputs) - 1

    def get_first_element(self, chunk):
        x = chunk * self.col_size + self.range[0]
        return felt(x), felt(self.f(x))

    def columns(self, n, usable):
        xs = list(range(self.range[0], self.largest() + 1))
        return _table_columns(n, usable, self.col_size, self.selector_constructor, [xs, [self.f(x) for x in xs]])


def _table_columns(n, usable, col_size, sc, value_lists):
    """value_lists[t][i] -> per list, one assignment (n ints mod R) per table column.  halo2's SimpleTableLayouter fills the
    unassigned tail of every table column, up to the last usable row, with its FIRST row ("default value"), which is why a
    partially filled last column still only holds table values (seen in the fixture pk.key: rows 16..57 of the third
    (0,127) column repeat row 0); the blinding rows stay 0."""
    out = []
    for vals in value_lists:
        ncols = -(-len(vals) // col_size)
        cols = []
        for c in range(max(ncols, sc.degree)):
            mult = sc.get_selector_val_at_idx(c)
            chunk = [felt(v) * mult % R for v in vals[c * col_size:(c + 1) * col_size]]
            cols.append(chunk)
        out.append(cols)
    # default value = first assigned row of each column; every column of ONE table region must have the same length
    res = []
    for cols in out:
        res.append([chunk + [chunk[0]] * (usable - len(chunk)) + [0] * (n - usable) if chunk else [0] * n for chunk in cols])
    return res


class BaseConfig:
    """src/circuit/ops/chip.rs:277-448"""

    def __init__(self, cs, inputs, output):
        self.cs, self.inputs, self.output = cs, inputs, output
        nonaccum, accum = {}, {}
        for i in range(output.num_blocks()):
            for j in range(output.num_inner_cols):
                nonaccum[(ADD, i, j)] = cs.selector()
                nonaccum[(SUB, i, j)] = cs.selector()
                nonaccum[(MULT, i, j)] = cs.selector()
        for i in range(output.num_blocks()):
            accum[(DOTINIT, i, 0)] = cs.selector()
            accum[(DOT, i, 0)] = cs.selector()
            accum[(CUMPROD, i, 0)] = cs.selector()
            accum[(CUMPRODINIT, i, 0)] = cs.selector()
            accum[(SUM, i, 0)] = cs.selector()
            accum[(SUMINIT, i, 0)] = cs.selector()
        for (op, blk, col) in sorted(nonaccum):
            s = nonaccum[(op, blk, col)]
            def gate(meta, op=op, blk=blk, col=col, s=s):
                sel = meta.query_selector(s)
                qis = [P.const(0), P.const(0)]
                for i in range(2 - NUM_INPUTS[op], 2):
                    qis[i] = inputs[i].query_rng(meta, blk, col, 0, 1)[0]
                off, rng = OFFSET_RNG[op]
                expected = output.query_rng(meta, blk, col, off, rng)
                return sel, [expected[rng - 1] - nonaccum_f(op, qis[0], qis[1])]
            cs.create_gate(OP_NAME[op], gate)
        for (op, blk, _c) in sorted(accum):
            s = accum[(op, blk, 0)]
            def gate(meta, op=op, blk=blk, s=s):
                sel = meta.query_selector(s)
                qis = [[], []]
                for i in range(2 - NUM_INPUTS[op], 2):
                    qis[i] = inputs[i].query_whole_block(meta, blk, 0, 1)
                off, rng = OFFSET_RNG[op]
                expected = output.query_rng(meta, blk, 0, off, rng)
                return sel, [expected[rng - 1] - accum_f(op, expected[0], qis[0], qis[1])]
            cs.create_gate(OP_NAME[op], gate)
        self.selectors = dict(nonaccum)
        self.selectors.update(accum)
        self.static_tables, self.static_selectors = {}, {}
        self.shared_table_inputs = []
        self.range_checks, self.range_selectors = {}, {}
        self.dynamic_lookup_selectors, self.dynamic_table_selectors = {}, []
        self.shuffle_input_selectors, self.shuffle_output_selectors = {}, []
        self.einsums = None

    def _synthetic(self, meta, length, index, x, y):
        return P.const(1) if length == 1 else meta.query_advice(index.inner[x][y], 0)

    def _sel_range_gate(self, length, index, x, y, multi_col_selector):
        def gate(meta):
            synthetic = self._synthetic(meta, length, index, x, y)
            if length == 1:
                e = P.const(0)
            else:
                e = P.const(1)
                for i in range(length):
                    e = e * (synthetic - P.const(i))
            return meta.query_selector(multi_col_selector), [e]
        self.cs.create_gate("range_check_on_sel", gate)

    def configure_lookup(self, input, output, index, lookup_range, logrows, name, f):
        """a static lookup table for the nonlinearity `name` (f: int -> int on the quantised domain)"""
        cs = self.cs
        if name in self.static_tables:
            return
        table = Table(cs, lookup_range, logrows, f, self.shared_table_inputs)
        self.static_tables[name] = table
        length = table.selector_constructor.degree
        for x in range(input.num_blocks()):
            for y in range(input.num_inner_cols):
                sel_ = cs.complex_selector()
                for col_idx, (icol, ocol) in enumerate(zip(table.table_inputs, table.table_outputs)):
                    def lk(meta, col_idx=col_idx, icol=icol, ocol=ocol):
                        sel = meta.query_selector(sel_)
                        synthetic = self._synthetic(meta, length, index, x, y)
                        iq = meta.query_advice(input.inner[x][y], 0)
                        oq = meta.query_advice(output.inner[x][y], 0)
                        col_expr = sel * table.selector_constructor.get_expr_at_idx(col_idx, synthetic)
                        not_expr = P.const(table.selector_constructor.get_selector_val_at_idx(col_idx)) - col_expr
                        dx, dy = table.get_first_element(col_idx)
                        return [(col_expr * iq + not_expr * P.const(dx), icol), (col_expr * oq + not_expr * P.const(dy), ocol)]
                    cs.lookup("", lk)
                self._sel_range_gate(length, index, x, y, sel_)
                self.static_selectors[(name, x, y)] = sel_

    def configure_range_check(self, input, index, rng, logrows):
        cs = self.cs
        rng = tuple(rng)
        if rng in self.range_checks:
            return
        rc = RangeCheck(cs, rng, logrows)
        self.range_checks[rng] = rc
        length = rc.selector_constructor.degree
        for x in range(input.num_blocks()):
            for y in range(input.num_inner_cols):
                sel_ = cs.complex_selector()
                for col_idx, icol in enumerate(rc.inputs):
                    def lk(meta, col_idx=col_idx, icol=icol):
                        sel = meta.query_selector(sel_)
                        synthetic = self._synthetic(meta, length, index, x, y)
                        iq = meta.query_advice(input.inner[x][y], 0)
                        dx = rc.get_first_element(col_idx)
                        col_expr = sel * rc.selector_constructor.get_expr_at_idx(col_idx, synthetic)
                        not_expr = P.const(rc.selector_constructor.get_selector_val_at_idx(col_idx)) - col_expr
                        return [(col_expr * iq + not_expr * P.const(dx), icol)]
                    cs.lookup("", lk)
                self._sel_range_gate(length, index, x, y, sel_)
                self.range_selectors[(rng, x, y)] = sel_

    def _configure_any(self, lookups, tables, name, in_sel, tab_sel):
        cs, one = self.cs, P.const(1)
        for q in range(tables[0].num_blocks()):
            s_table = cs.complex_selector()
            for x in range(lookups[0].num_blocks()):
                for y in range(lookups[0].num_inner_cols):
                    s_lookup = cs.complex_selector()
                    def lk(meta, q=q, x=x, y=y, s_lookup=s_lookup):
                        sl = meta.query_selector(s_lookup)
                        st = meta.query_selector(s_table)
                        lq = [one] + [meta.query_advice(l.inner[x][y], 0) for l in lookups]
                        tq = [one] + [meta.query_advice(t.inner[q][0], 0) for t in tables]
                        return [(a * sl, b * st) for a, b in zip(lq, tq)]
                    cs.lookup_any(name, lk)
                    in_sel.setdefault((q, (x, y)), s_lookup)
            tab_sel.append(s_table)

    def configure_dynamic_lookup(self, lookups, tables):
        self._configure_any(lookups, tables, "lookup", self.dynamic_lookup_selectors, self.dynamic_table_selectors)

    def configure_shuffles(self, inputs, outputs):
        self._configure_any(inputs, outputs, "shuffle", self.shuffle_input_selectors, self.shuffle_output_selectors)

    def configure_einsums(self, reduction_length, max_num_output_axes, num_inner_cols, logrows):
        self.einsums = Einsums(self.cs, reduction_length, max_num_output_axes, num_inner_cols, logrows)


# InputPhases in declaration order (src/circuit/ops/chip/einsum/mod.rs:423-429)
FIRST_PHASE, SECOND_PHASE, BOTH_FIRST, MIXED, BOTH_SECOND = range(5)


class Einsums:
    """Freivalds einsum columns and gates (src/circuit/ops/chip/einsum/mod.rs:60-94 configure_universal, :487-684
    ContractionConfig::new, :715-783 RLCConfig::new)"""

    def __init__(self, cs, reduction_length, max_num_output_axes, num_inner_cols, logrows):
        cap = reduction_length
        self.inputs = [VarTensor.new_advice(cs, logrows, num_inner_cols, cap), VarTensor.new_advice(cs, logrows, num_inner_cols, cap),
                       VarTensor.new_advice_in_second_phase(cs, logrows, num_inner_cols, cap),
                       VarTensor.new_advice_in_second_phase(cs, logrows, num_inner_cols, cap)]
        self.outputs = [VarTensor.new_advice(cs, logrows, num_inner_cols, cap), VarTensor.new_advice_in_second_phase(cs, logrows, num_inner_cols, cap)]
        self.contraction_selectors = self._contraction(cs, [[self.inputs[0], self.inputs[1]], [self.inputs[2], self.inputs[3]]], self.outputs)
        self.rlc = [self._rlc(cs, [self.inputs[0], self.inputs[2]], self.outputs[1]) for _ in range(max_num_output_axes)]

    @staticmethod
    def _contraction(cs, inputs, outputs):
        selectors = {}
        num_blocks, width = outputs[0].num_blocks(), outputs[0].num_inner_cols
        for ph in (BOTH_FIRST, MIXED, BOTH_SECOND):
            for i in range(num_blocks):
                for j in range(width):
                    selectors[((MULT, ph), i, j)] = cs.selector()
                for i2 in range(num_blocks):              # (sic) nested in the block loop upstream: re-inserted keys leave orphan selectors
                    selectors[((DOTINIT, ph), i2, 0)] = cs.selector()
                    selectors[((DOT, ph), i2, 0)] = cs.selector()
        for ph in (FIRST_PHASE, SECOND_PHASE):
            for i in range(num_blocks):
                selectors[((SUMINIT, ph), i, 0)] = cs.selector()
                selectors[((SUM, ph), i, 0)] = cs.selector()
                selectors[((CUMPRODINIT, ph), i, 0)] = cs.selector()
                selectors[((CUMPROD, ph), i, 0)] = cs.selector()
        for key in sorted(selectors):
            (op, ph), blk, col = key
            s = selectors[key]
            ins = {FIRST_PHASE: [inputs[0][0]], SECOND_PHASE: [inputs[1][0]], BOTH_FIRST: [inputs[0][0], inputs[0][1]],
                   MIXED: [inputs[0][0], inputs[1][0]], BOTH_SECOND: [inputs[1][0], inputs[1][1]]}[ph]
            out = outputs[0] if ph in (FIRST_PHASE, BOTH_FIRST) else outputs[1]
            assert len(ins) == NUM_INPUTS[op]
            if op == MULT:
                def gate(meta, s=s, ins=ins, out=out, blk=blk, col=col, op=op):
                    sel = meta.query_selector(s)
                    qis = [P.const(0), P.const(0)]
                    for t, inp in enumerate(ins):
                        qis[t] = inp.query_rng(meta, blk, col, 0, 1)[0]
                    off, rng = OFFSET_RNG[op]
                    expected = out.query_rng(meta, blk, col, off, rng)
                    return sel, [expected[rng - 1] - nonaccum_f(op, qis[0], qis[1])]
            else:
                def gate(meta, s=s, ins=ins, out=out, blk=blk, op=op):
                    sel = meta.query_selector(s)
                    qis = [[], []]
                    for t, inp in enumerate(ins):
                        qis[t] = inp.query_whole_block(meta, blk, 0, 1)
                    off, rng = OFFSET_RNG[op]
                    expected = out.query_rng(meta, blk, 0, off, rng)
                    return sel, [expected[rng - 1] - accum_f(op, expected[0], qis[1], qis[0])]
            cs.create_gate(OP_NAME[op], gate)
        return selectors

    @staticmethod
    def _rlc(cs, inputs, output):
        challenge = cs.challenge_usable_after(0)
        selectors = {}
        for phase, inp in enumerate(inputs):
            for blk in range(inp.num_blocks()):
                selectors[(phase, blk)] = (cs.selector(), cs.selector())
        width = output.num_inner_cols
        powers, rp = [], P.const(1)
        for _ in range(width):
            rp = rp * P.chal(challenge)
            powers.append(rp)
        for (phase, blk) in sorted(selectors):
            init_s, acc_s = selectors[(phase, blk)]
            def g_init(meta, phase=phase, blk=blk, init_s=init_s):
                sel = meta.query_selector(init_s)
                ie = inputs[phase].query_whole_block(meta, blk, 0, 1)
                expected = output.query_rng(meta, blk, 0, 0, 1)
                return sel, [expected[0] - accum_f(DOT, P.const(0), powers[::-1], ie)]
            cs.create_gate("init", g_init)
            def g_acc(meta, phase=phase, blk=blk, acc_s=acc_s):
                sel = meta.query_selector(acc_s)
                ie = inputs[phase].query_whole_block(meta, blk, 0, 1)
                expected = output.query_rng(meta, blk, 0, -1, 2)
                return sel, [expected[1] - accum_f(DOT, expected[0] * powers[-1], powers[::-1], ie)]
            cs.create_gate("acc", g_acc)
        return dict(challenge=challenge, selectors=selectors)


class GraphSettings:
    """the fields of ezkl's settings.json that shape the constraint system (src/graph/mod.rs:453-492)"""

    def __init__(self, logrows, num_inner_cols, total_assignments, total_const_size=0, required_range_checks=(), required_lookups=(),
                 lookup_range=(0, 0), model_instance_shapes=(), total_dynamic_col_size=0, num_dynamic_lookups=0, total_shuffle_col_size=0,
                 num_shuffles=0, einsum_reduction_length=0, einsum_max_output_axes=0):
        self.logrows, self.num_inner_cols, self.total_assignments, self.total_const_size = logrows, num_inner_cols, total_assignments, total_const_size
        self.required_range_checks = [tuple(r) for r in required_range_checks]
        self.required_lookups = list(required_lookups)          # [(name, f)]
        self.lookup_range = tuple(lookup_range)
        self.model_instance_shapes = [list(s) for s in model_instance_shapes]
        self.total_dynamic_col_size, self.num_dynamic_lookups = total_dynamic_col_size, num_dynamic_lookups
        self.total_shuffle_col_size, self.num_shuffles = total_shuffle_col_size, num_shuffles
        self.einsum_reduction_length, self.einsum_max_output_axes = einsum_reduction_length, einsum_max_output_axes

    @classmethod
    def from_json(cls, j):
        ra = j["run_args"]
        assert not j.get("required_lookups"), "static lookups need their nonlinearity: pass required_lookups=[(name, f)] explicitly"
        ep = j.get("einsum_params", {})
        assert not ep.get("total_einsum_col_size", 0), "einsum parameters: pass einsum_reduction_length / einsum_max_output_axes explicitly"
        return cls(ra["logrows"], ra["num_inner_cols"], j["total_assignments"], j["total_const_size"], j["required_range_checks"], (),
                   ra.get("lookup_range", (0, 0)), j.get("model_instance_shapes", ()), j.get("total_dynamic_col_size", 0),
                   j.get("num_dynamic_lookups", 0), j.get("total_shuffle_col_size", 0), j.get("num_shuffles", 0))

    def requires_dynamic_lookup(self): return self.num_dynamic_lookups > 0
    def requires_shuffle(self): return self.num_shuffles > 0
    def dynamic_lookup_and_shuffle_col_size(self): return self.total_dynamic_col_size + self.total_shuffle_col_size


class GraphConfig:
    """GraphCircuit::configure_with_params (src/graph/mod.rs:1945-2004): ModelVars::new -> instance column -> Model::configure"""

    def __init__(self, settings):
        s = self.settings = settings
        cs = self.cs = ConstraintSystem()
        k, w = s.logrows, s.num_inner_cols
        # ModelVars::new
        self.advices = [VarTensor.new_advice(cs, k, w, s.total_assignments) for _ in range(3)]
        if s.requires_dynamic_lookup() or s.requires_shuffle():
            for _ in range(3):
                self.advices.append(VarTensor.new_advice(cs, k, 1, s.dynamic_lookup_and_shuffle_col_size()))
        self.const_cols = VarTensor.constant_cols(cs, k, s.total_const_size)
        # instantiate_instance: one instance column for all public tensors
        self.instance = None
        if s.model_instance_shapes:
            self.instance = cs.instance_column()
            cs.enable_equality(self.instance)
        # Model::configure
        base = self.base = BaseConfig(cs, self.advices[0:2], self.advices[2])
        inp, index, out = self.advices[0], self.advices[1], self.advices[2]
        for name, f in s.required_lookups:
            base.configure_lookup(inp, out, index, s.lookup_range, k, name, f)
        for rng in s.required_range_checks:
            base.configure_range_check(inp, index, rng, k)
        if s.requires_dynamic_lookup():
            base.configure_dynamic_lookup(self.advices[0:3], self.advices[3:6])
        if s.requires_shuffle():
            base.configure_shuffles(self.advices[0:3], self.advices[3:6])
        if s.einsum_reduction_length > 0:
            base.configure_einsums(s.einsum_reduction_length, s.einsum_max_output_axes, w, k)
        # create_domain / keygen / create_proof all run chunk_lookups() right after configure
        cs.chunk_lookups()

    def table_columns(self):
        """{fixed column index: assignment (n ints)} of every lookup-table column (layout_tables / layout_range_checks)"""
        n, out = 1 << self.settings.logrows, {}
        usable = n - self.cs.blinding_factors() - 1
        first = True
        for name, t in self.base.static_tables.items():
            ins, outs = t.columns(n, usable)
            if first:
                for col, a in zip(t.table_inputs, ins): out[col.index] = a
            for col, a in zip(t.table_outputs, outs): out[col.index] = a
            first = False
        for rc in self.base.range_checks.values():
            for col, a in zip(rc.inputs, rc.columns(n, usable)): out[col.index] = a
        return out
