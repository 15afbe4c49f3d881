"""Front-end KATs restated from reference src/tests/test_grammar.py:24-137 (same
program text, same expected strings; built with this repo's own IR)."""
import pytest

from soda_amd import grammar, ir, util


KITCHEN_SINK = r'''
border: ignore
burst width: 512
cluster: none
iterate: 2
kernel: name
unroll factor: 1
input dram 0 float: bbb
input dram 1 uint6: a(233, *)
param int8: p0
param int9, dup 3: p1[23]
param int10, partition complete: p2[23]
param int11, partition complete dim=1: p2[23]
param int12, partition cyclic factor=23: p3[233]
param int13, partition cyclic factor=23 dim=2: p4[233][233]
param int14, partition complete, dup 3: p5[23]
local int27:
  int32 l = int32(a(0, 0) ~1 + b(1, 0))
  int32 g = int32(a(0, 0) ~1 + p0 + p1[1][3])
  c(0, 0) ~3 = +-+-l * --+~l
output dram 2 double:
  float18_3 l = float18_3(c(0, 1) ~5) + a(1, 0)
  d(0, 0) = sqrt(float15(l <= (l / 2)))
output dram 3 double:
  float18_3 l = float18_3(c(0, 1) ~5) + a(1, 0)
  e(0, 0) = float15(l + (l / 2))
'''.strip('\n')


def test_syntax_round_trip():
  """reference test_grammar.py:24-61: str(model) == source, byte for byte."""
  assert str(grammar.parse(KITCHEN_SINK)) == KITCHEN_SINK


@pytest.fixture
def parts():
  int8 = ir.Type('int8')
  ref = ir.Ref('foo', (0, 23))
  expr = ir.Ref('bar', (233, 42), haoda_type=int8)
  let = ir.Let(int8, 'foo_l', ir.Ref('bar_l', (42, 2333)))
  let2 = ir.Let(int8, 'foo_l2', ir.Ref('bar_l2', (0, 42)))
  return int8, ref, expr, let, let2


def test_input(parts):
  int8 = parts[0]
  assert str(grammar.InputStmt(haoda_type=int8, name='foo', tile_size=[],
                               dram=())) == 'input dram 0 int8: foo'
  assert str(grammar.InputStmt(haoda_type=int8, name='foo', tile_size=[23],
                               dram=())) == 'input dram 0 int8: foo(23, *)'
  assert str(grammar.InputStmt(
      haoda_type=int8, name='foo', tile_size=[23, 233],
      dram=())) == 'input dram 0 int8: foo(23, 233, *)'


@pytest.mark.parametrize('cls,prefix', [(grammar.LocalStmt, 'local'),
                                        (grammar.OutputStmt, 'output dram 0')])
def test_local_and_output(parts, cls, prefix):
  int8, ref, expr, let, let2 = parts
  assert str(cls(haoda_type=int8, let=[], ref=ref, expr=expr,
                 dram=())) == prefix + ' int8: foo(0, 23) = bar(233, 42)'
  assert str(cls(haoda_type=int8, let=[let], ref=ref, expr=expr, dram=())) == (
      prefix + ' int8:\n  int8 foo_l = bar_l(42, 2333)\n'
      '  foo(0, 23) = bar(233, 42)')
  assert str(cls(haoda_type=int8, let=[let, let2], ref=ref, expr=expr,
                 dram=())) == (
      prefix + ' int8:\n  int8 foo_l = bar_l(42, 2333)\n'
      '  int8 foo_l2 = bar_l2(0, 42)\n  foo(0, 23) = bar(233, 42)')


def test_precedence_and_association():
  e = grammar.parse_expr('a(0) + b(0) * c(0) - d(0) / 2 % 3')
  assert str(e) == 'a(0) + (b(0) * c(0)) - (d(0) / 2 % 3)'
  e = grammar.parse_expr('a(0) - (b(0) - c(0))')
  assert str(e) == 'a(0) - (b(0) - c(0))'
  e = grammar.parse_expr('x < y == z & 1 | 2 ^ 3 && p || q')
  assert str(e) == '(((((x < y) == z) & 1) | (2 ^ 3)) && p) || q'
  # C text groups exactly as the DSL does
  assert ir.c_expr(grammar.parse_expr('a(0) - (b(0) - c(0)) * 2'),
                   lambda r: r.name) == '(a - ((b - c) * 2))'


def test_literals_and_calls():
  e = grammar.parse_expr('.125f * in(1, 0, 0) + 0.2f + 2.0 + 1e-3 + 65535 + -106')
  nums = [n for n in e.walk() if isinstance(n, ir.Num)]
  assert [n.literal for n in nums] == ['.125f', '0.2f', '2.0', '1e-3', '65535',
                                       '106']
  assert str(nums[0].haoda_type) == 'float'
  assert str(nums[2].haoda_type) == 'double'   # reference README.md:222
  assert str(nums[4].haoda_type) == 'int32'
  e = grammar.parse_expr('min(a(0, 0), b(0, 1), 3) + sqrt(1.0f + c(0, 0))')
  calls = [n for n in e.walk() if isinstance(n, ir.Call)]
  assert [c.name for c in calls] == ['min', 'sqrt']
  assert len(calls[0].args) == 3


def test_statement_order_is_free_and_comments():
  p = grammar.parse('''
    # a comment
    iterate: 1
    output float: o(0, 0) = i(0, 0) + 1.0f   # trailing
    input float: i(16, *)
    unroll factor: 2
    kernel: k
    burst width: 64
  ''')
  assert p.app_name == 'k' and p.dim == 2 and p.tile_size == (16, 0)
  assert p.border is None and p.cluster is None


def test_multi_bank_dram():
  p = grammar.parse('kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
                    'input dram 0.1 float: i(8, *)\n'
                    'output dram 2.3 float: o(0, 0) = i(0, 0)\n')
  assert p.input_stmts[0].dram == (0, 1)
  assert p.output_stmts[0].dram == (2, 3)
  assert str(p.input_stmts[0]) == 'input dram 0.1 float: i(8, *)'


@pytest.mark.parametrize('text,needle', [
    ('kernel: k\nburst width: 64\nunroll factor: 1\ninput float: i\n'
     'output float: o(0) = i(0)', 'iterate'),
    ('kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
     'output float: o(0) = i(0)', 'input'),
    ('kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
     'input float: i\noutput float: o(0) = i(0) +', 'operand'),
    ('kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
     'input float: i\noutput float: o(0) = i(0) $', 'unexpected character'),
    ('kernel: k\nkernel: j\nburst width: 64\nunroll factor: 1\niterate: 1\n'
     'input float: i\noutput float: o(0) = i(0)', 'duplicate'),
    ('kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
     'border: maybe\ninput float: i\noutput float: o(0) = i(0)', 'ignore'),
])
def test_syntax_errors(text, needle):
  with pytest.raises(util.SodaSyntaxError) as e:
    grammar.parse(text)
  assert needle in str(e.value)


def test_tile_size_mismatch():
  with pytest.raises(util.SemanticError) as e:
    grammar.parse('kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
                  'input float: a(8, *)\ninput float: b(16, *)\n'
                  'output float: o(0, 0) = a(0, 0) + b(0, 0)')
  assert "doesn't match previous one" in str(e.value)


def test_types():
  t = ir.Type('uint16')
  assert (t.c_type, t.np_name, t.size_in_bytes, t.is_float) == (
      'uint16_t', 'uint16', 2, False)
  assert ir.Type('float').c_type == 'float'
  assert ir.Type('double').np_name == 'float64'
  assert not ir.Type('uint6').is_native and not ir.Type('float18_3').is_native
  with pytest.raises(util.SemanticError):
    ir.Type('uint6').c_type
  with pytest.raises(util.SemanticError):
    ir.Type('quad')
