"""Dependency-free front-end for the .soda DSL: tokenizer + recursive descent.

Accepts the language of reference src/soda/grammar.py:15-46 (program layout,
directives, input/param/local/output statements) plus the expression grammar
the reference pulls from haoda (`ir.GRAMMAR`, grammar.py:46; class order at
grammar.py:209-232).  The reference builds this with textX; neither textX nor
haoda exists here, so the parser is hand-written.  `str()` of every statement
and of the whole program reproduces the normal form pinned by the reference's
unit tests (src/tests/test_grammar.py:24-137).
"""
import re
from typing import Dict, List, Optional, Sequence, Tuple

from soda_amd import ir, util

# --------------------------------------------------------------------------
# tokens
# --------------------------------------------------------------------------

_TOKEN_RE = re.compile(
    r'''
    (?P<ws>[ \t\r\n]+)
  | (?P<comment>\#[^\n]*)
  | (?P<float>(?:(?:\d*\.\d+|\d+\.)(?:[Ee][+-]?\d+)?|\d+[Ee][+-]?\d+)[FfLl]?|\d+[Ff])
  | (?P<int>0[Xx][0-9a-fA-F]+[UuLl]*|0[Bb][01]+[UuLl]*|\d+[UuLl]*)
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>\|\||&&|==|!=|<=|>=|[-+*/%~!|^&<>()\[\],:.=])
''', re.VERBOSE)

_TYPE_RE = re.compile(
    r'^(u?int[1-9]\d*(_[1-9]\d*)?|float[1-9]\d*(_[1-9]\d*)?|float|double|half)$'
)


class Token:
  __slots__ = ('kind', 'text', 'line', 'col')

  def __init__(self, kind, text, line, col):
    self.kind = kind
    self.text = text
    self.line = line
    self.col = col

  def __repr__(self):
    return '%s(%r)@%d:%d' % (self.kind, self.text, self.line, self.col)


def tokenize(text: str) -> List[Token]:
  tokens = []
  pos = 0
  line = 1
  line_start = 0
  while pos < len(text):
    m = _TOKEN_RE.match(text, pos)
    if m is None:
      raise util.SodaSyntaxError('unexpected character %r' % text[pos], line,
                                 pos - line_start + 1)
    kind = m.lastgroup
    tok = m.group(kind)
    if kind not in ('ws', 'comment'):
      tokens.append(Token(kind, tok, line, pos - line_start + 1))
    newlines = tok.count('\n')
    if newlines:
      line += newlines
      line_start = pos + tok.rfind('\n') + 1
    pos = m.end()
  tokens.append(Token('eof', '', line, pos - line_start + 1))
  return tokens


# --------------------------------------------------------------------------
# statements
# --------------------------------------------------------------------------

class InputStmt:
  """`input dram 1 float: u(32, *)` -- a tiled input tensor.

  `tile_size` always ends with the unbounded dimension's 0, as in the
  reference (grammar.py:58-65)."""

  def __init__(self, haoda_type, name: str, tile_size: Sequence[int] = (),
               dram: Sequence[int] = ()):
    self.haoda_type = ir.Type(str(haoda_type))
    self.name = name
    self.dram = tuple(dram) or (0,)
    self.tile_size = tuple(tile_size) + (0,)

  def __str__(self) -> str:
    out = 'input dram %s %s: %s' % ('.'.join(map(str, self.dram)),
                                    self.haoda_type, self.name)
    if self.tile_size[:-1]:
      out += '(%s, *)' % ', '.join(map(str, self.tile_size[:-1]))
    return out


class _ComputeStmt:
  """Common part of `local` and `output` statements."""

  keyword = ''

  def __init__(self, haoda_type, ref: ir.Ref, expr: ir.Node,
               let: Sequence[ir.Let] = (), dram: Sequence[int] = ()):
    self.haoda_type = ir.Type(str(haoda_type))
    self.ref = ref
    self.expr = expr
    self.let = tuple(let)
    self.dram = tuple(dram) or (0,)
    let_types = {l.name: l.haoda_type for l in self.let}

    def tag(n):
      if isinstance(n, ir.Var) and n.haoda_type is None and n.name in let_types:
        n.haoda_type = let_types[n.name]
      return n

    self.expr = self.expr.transform(tag)
    self.let = tuple(l.transform(tag) for l in self.let)

  @property
  def name(self) -> str:
    return self.ref.name

  def _body(self) -> str:
    lets = ''
    if self.let:
      lets = '\n  %s\n ' % '\n  '.join(map(str, self.let))
    return '%s:%s %s = %s' % (self.haoda_type, lets, self.ref, self.expr)

  def propagate_type(self, symbol_table: Dict[str, ir.Type]) -> None:
    """Tag references with their tensor's type and wrap the expression in a
    cast to the statement type when they differ (ref grammar.py:123-136)."""
    table = dict(symbol_table)

    def tag(n):
      if isinstance(n, (ir.Ref, ir.Var)) and n.haoda_type is None:
        if n.name in table:
          n.haoda_type = table[n.name]
      return n

    new_lets = []
    for l in self.let:
      l = l.transform(tag)
      t = l.haoda_type or l.expr.haoda_type
      if l.haoda_type is None:
        l = ir.Let(t, l.name, l.expr)
      table[l.name] = t
      new_lets.append(l)
    self.let = tuple(new_lets)
    self.expr = self.expr.transform(tag)
    if isinstance(self.expr, ir.Cast) and self.expr.haoda_type == self.haoda_type:
      return
    if self.expr.haoda_type != self.haoda_type:
      self.expr = ir.Cast(self.haoda_type, self.expr)


class LocalStmt(_ComputeStmt):
  keyword = 'local'

  def __str__(self) -> str:
    return 'local ' + self._body()


class OutputStmt(_ComputeStmt):
  keyword = 'output'

  def __str__(self) -> str:
    return 'output dram %s %s' % ('.'.join(map(str, self.dram)), self._body())


class ParamAttr:

  def __init__(self, dup: Optional[int] = None, strategy: Optional[str] = None,
               factor: Optional[int] = None, dim: Optional[int] = None):
    self.dup = dup
    self.strategy = strategy
    self.factor = factor
    self.dim = dim

  def __str__(self) -> str:
    if self.dup is not None:
      return 'dup %d' % self.dup
    out = 'partition %s' % self.strategy
    if self.strategy == 'cyclic':
      out += ' factor=%d' % self.factor
    if self.dim is not None:
      out += ' dim=%d' % self.dim
    return out


class ParamStmt:

  def __init__(self, haoda_type, name: str, attr: Sequence[ParamAttr] = (),
               size: Sequence[int] = (), dram: Sequence[int] = ()):
    self.haoda_type = ir.Type(str(haoda_type))
    self.name = name
    self.attr = tuple(attr)
    self.size = tuple(size)
    self.dram = tuple(dram)

  def __str__(self) -> str:
    return 'param %s%s: %s%s' % (self.haoda_type, ''.join(
        ', %s' % a for a in self.attr), self.name, ''.join(
            '[%d]' % s for s in self.size))


class SodaProgram:
  """Parse result; field names follow reference grammar.py:173-207."""

  def __init__(self, *, border, burst_width, cluster, iterate, app_name,
               unroll_factor, input_stmts, param_stmts, local_stmts,
               output_stmts):
    self.border = border
    self.burst_width = burst_width
    self.cluster = cluster
    self.iterate = iterate
    self.app_name = app_name
    self.unroll_factor = unroll_factor
    self.input_stmts = list(input_stmts)
    self.param_stmts = list(param_stmts)
    self.local_stmts = list(local_stmts)
    self.output_stmts = list(output_stmts)
    tile_size = None
    for stmt in self.input_stmts:
      if tile_size is not None:
        if stmt.tile_size[:-1] and stmt.tile_size != tile_size:
          raise util.SemanticError(
              "tile size %s doesn't match previous one %s" %
              (stmt.tile_size, tile_size))
      elif stmt.tile_size[:-1]:
        tile_size = stmt.tile_size
    if tile_size is None:  # 1-D program
      tile_size = self.input_stmts[-1].tile_size
    self.tile_size = tile_size
    self.dim = len(tile_size)

  def __str__(self) -> str:
    lines = [
        'border: %s' % self.border if self.border is not None else '',
        'burst width: %s' % self.burst_width,
        'cluster: %s' % self.cluster if self.cluster is not None else '',
        'iterate: %s' % self.iterate,
        'kernel: %s' % self.app_name,
        'unroll factor: %s' % self.unroll_factor,
    ]
    for group in (self.input_stmts, self.param_stmts, self.local_stmts,
                  self.output_stmts):
      lines.extend(map(str, group))
    return '\n'.join(l for l in lines if l)


# --------------------------------------------------------------------------
# parser
# --------------------------------------------------------------------------

_STMT_KEYWORDS = ('input', 'output', 'local', 'param', 'kernel', 'burst',
                  'unroll', 'iterate', 'border', 'cluster')


class _Parser:

  def __init__(self, text: str):
    self.toks = tokenize(text)
    self.i = 0

  # -- helpers -------------------------------------------------------------
  @property
  def tok(self) -> Token:
    return self.toks[self.i]

  def peek(self, k: int = 1) -> Token:
    return self.toks[min(self.i + k, len(self.toks) - 1)]

  def error(self, what: str, tok: Optional[Token] = None):
    tok = tok or self.tok
    found = repr(tok.text) if tok.kind != 'eof' else 'end of input'
    raise util.SodaSyntaxError('expected %s, found %s' % (what, found),
                               tok.line, tok.col)

  def at(self, text: str) -> bool:
    return self.tok.text == text and self.tok.kind in ('op', 'id')

  def accept(self, text: str) -> bool:
    if self.at(text):
      self.i += 1
      return True
    return False

  def expect(self, text: str) -> Token:
    if not self.at(text):
      self.error("'%s'" % text)
    self.i += 1
    return self.toks[self.i - 1]

  def ident(self, what: str = 'identifier') -> str:
    if self.tok.kind != 'id':
      self.error(what)
    self.i += 1
    return self.toks[self.i - 1].text

  def uint(self, what: str = 'integer') -> int:
    if self.tok.kind != 'int' or not self.tok.text.isdigit():
      self.error(what)
    self.i += 1
    return int(self.toks[self.i - 1].text)

  def sint(self, what: str = 'integer') -> int:
    sign = 1
    if self.at('-'):
      self.i += 1
      sign = -1
    elif self.at('+'):
      self.i += 1
    return sign * self.uint(what)

  def at_type(self) -> bool:
    return self.tok.kind == 'id' and bool(_TYPE_RE.match(self.tok.text))

  def type_(self) -> ir.Type:
    if not self.at_type():
      self.error('a type')
    self.i += 1
    return ir.Type(self.toks[self.i - 1].text)

  # -- program -------------------------------------------------------------
  def program(self) -> SodaProgram:
    fields = dict(border=None, burst_width=None, cluster=None, iterate=None,
                  app_name=None, unroll_factor=None)
    inputs, params, locals_, outputs = [], [], [], []

    def set_once(key, value, tok):
      if fields[key] is not None:
        raise util.SodaSyntaxError('duplicate `%s` directive' %
                                   key.replace('_', ' '), tok.line, tok.col)
      fields[key] = value

    while self.tok.kind != 'eof':
      tok = self.tok
      if tok.kind != 'id' or tok.text not in _STMT_KEYWORDS:
        self.error('a directive or statement')
      self.i += 1
      kw = tok.text
      if kw == 'kernel':
        self.expect(':')
        set_once('app_name', self.ident('kernel name'), tok)
      elif kw == 'burst':
        self.expect('width')
        self.expect(':')
        set_once('burst_width', self.uint(), tok)
      elif kw == 'unroll':
        self.expect('factor')
        self.expect(':')
        set_once('unroll_factor', self.uint(), tok)
      elif kw == 'iterate':
        self.expect(':')
        set_once('iterate', self.uint(), tok)
      elif kw == 'border':
        self.expect(':')
        value = self.ident()
        if value not in ('ignore', 'preserve'):
          self.error("'ignore' or 'preserve'", self.toks[self.i - 1])
        set_once('border', value, tok)
      elif kw == 'cluster':
        self.expect(':')
        value = self.ident()
        if value not in ('none', 'fine', 'coarse', 'full'):
          self.error("'none', 'fine', 'coarse' or 'full'",
                     self.toks[self.i - 1])
        set_once('cluster', value, tok)
      elif kw == 'input':
        inputs.append(self.input_stmt())
      elif kw == 'param':
        params.append(self.param_stmt())
      elif kw == 'local':
        locals_.append(self.compute_stmt(LocalStmt))
      elif kw == 'output':
        outputs.append(self.compute_stmt(OutputStmt))
    for key, spelled in (('burst_width', 'burst width'), ('iterate', 'iterate'),
                         ('app_name', 'kernel'),
                         ('unroll_factor', 'unroll factor')):
      if fields[key] is None:
        self.error('`%s:` directive' % spelled)
    if not inputs:
      self.error('at least one input statement')
    if not outputs:
      self.error('at least one output statement')
    return SodaProgram(input_stmts=inputs, param_stmts=params,
                       local_stmts=locals_, output_stmts=outputs, **fields)

  def dram(self) -> Tuple[int, ...]:
    banks: List[int] = []
    if self.accept('dram'):
      # `dram 0.1` lexes as a float literal; split it back into banks
      if self.tok.kind == 'float' and re.match(r'^\d+\.\d+$', self.tok.text):
        banks.extend(int(b) for b in self.tok.text.split('.'))
        self.i += 1
      else:
        banks.append(self.uint('dram bank'))
      while self.at('.') or (self.tok.kind == 'float' and
                             re.match(r'^\.\d+$', self.tok.text)):
        if self.tok.kind == 'float':
          banks.append(int(self.tok.text[1:]))
          self.i += 1
        else:
          self.i += 1
          banks.append(self.uint('dram bank'))
    return tuple(banks)

  def input_stmt(self) -> InputStmt:
    dram = self.dram()
    haoda_type = self.type_()
    self.expect(':')
    name = self.ident('input name')
    tile_size: List[int] = []
    if self.accept('('):
      while not self.at('*'):
        tile_size.append(self.uint('tile size'))
        self.expect(',')
      self.expect('*')
      self.expect(')')
    return InputStmt(haoda_type, name, tile_size, dram)

  def param_stmt(self) -> ParamStmt:
    dram = self.dram()
    haoda_type = self.type_()
    attrs = []
    while self.accept(','):
      if self.accept('dup'):
        attrs.append(ParamAttr(dup=self.sint()))
      else:
        self.expect('partition')
        strategy = self.ident()
        factor = dim = None
        if strategy == 'cyclic':
          self.expect('factor')
          self.expect('=')
          factor = self.sint()
        elif strategy != 'complete':
          self.error("'complete' or 'cyclic'", self.toks[self.i - 1])
        if self.accept('dim'):
          self.expect('=')
          dim = self.sint()
        attrs.append(ParamAttr(strategy=strategy, factor=factor, dim=dim))
    self.expect(':')
    name = self.ident('param name')
    size = []
    while self.accept('['):
      size.append(self.uint())
      self.expect(']')
    return ParamStmt(haoda_type, name, attrs, size, dram)

  def compute_stmt(self, cls):
    dram = self.dram() if cls is OutputStmt else ()
    haoda_type = self.type_()
    self.expect(':')
    lets = []
    while True:
      # Let: [Type] ID '=' ...      Ref: ID '(' ...
      if self.at_type() and self.peek().kind == 'id':
        t = self.type_()
        name = self.ident()
        self.expect('=')
        lets.append(ir.Let(t, name, self.expr()))
      elif self.tok.kind == 'id' and self.peek().text == '=':
        name = self.ident()
        self.expect('=')
        lets.append(ir.Let(None, name, self.expr()))
      else:
        break
    ref = self.ref()
    self.expect('=')
    expr = self.expr()
    return cls(haoda_type, ref, expr, lets, dram)

  def ref(self) -> ir.Ref:
    name = self.ident('tensor name')
    self.expect('(')
    idx = [self.sint('index')]
    while self.accept(','):
      idx.append(self.sint('index'))
    self.expect(')')
    lat = None
    if self.accept('~'):
      lat = self.sint('latency')
    return ir.Ref(name, idx, lat)

  # -- expressions ---------------------------------------------------------
  def expr(self, level: int = 0) -> ir.Node:
    if level == len(ir.LEVELS):
      return self.unary()
    ops = ir.LEVELS[level][1]
    operands = [self.expr(level + 1)]
    operators = []
    while self.tok.kind == 'op' and self.tok.text in ops:
      # `a(0, 0) ~1` belongs to the Ref, and `=` never starts an operator here
      operators.append(self.tok.text)
      self.i += 1
      operands.append(self.expr(level + 1))
    if not operators:
      return operands[0]
    return ir.Chain(operands, operators)

  def unary(self) -> ir.Node:
    ops = []
    while self.tok.kind == 'op' and self.tok.text in ('+', '-', '~', '!'):
      ops.append(self.tok.text)
      self.i += 1
    operand = self.operand()
    if ops:
      return ir.Unary(ops, operand)
    return operand

  def operand(self) -> ir.Node:
    tok = self.tok
    if tok.kind in ('int', 'float'):
      self.i += 1
      return ir.Num(tok.text)
    if self.accept('('):
      inner = self.expr()
      self.expect(')')
      return inner
    if tok.kind == 'id':
      if self.at_type() and self.peek().text == '(':
        t = self.type_()
        self.expect('(')
        inner = self.expr()
        self.expect(')')
        return ir.Cast(t, inner)
      if self.peek().text == '(':
        if tok.text in ir.FUNC_NAMES:
          self.i += 2
          args = [self.expr()]
          while self.accept(','):
            args.append(self.expr())
          self.expect(')')
          return ir.Call(tok.text, args)
        return self.ref()
      if tok.text in _STMT_KEYWORDS:
        self.error('an operand')
      self.i += 1
      idx = []
      while self.at('[') :
        self.i += 1
        idx.append(self.sint())
        self.expect(']')
      return ir.Var(tok.text, idx)
    self.error('an operand')


def parse(text: str) -> SodaProgram:
  """Parses .soda source text (the `model_from_str` of reference sodac.py:144)."""
  return _Parser(text).program()


def parse_file(path: str) -> SodaProgram:
  with open(path) as f:
    return parse(f.read())


def parse_expr(text: str) -> ir.Node:
  p = _Parser(text)
  node = p.expr()
  if p.tok.kind != 'eof':
    p.error('end of expression')
  return node
