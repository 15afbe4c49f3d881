#!/usr/bin/env python3
"""Static check of hand-counted vector-memory waits in compiled gfx950 ISA.

The marching kernels issue their input-row loads as inline asm that hipcc does
not count, and wait for them with hand-counted `s_waitcnt vmcnt(N)`
(soda_amd/csrc/soda_rt.h, "loads the compiler does not count").  Two things
can go wrong silently: the count can be off (a row is used before it has
landed), and the compiler can touch a destination register while its load is
still in flight (a copy inserted by the register allocator reads stale data).
This tool replays the ISA text of a kernel (`hipcc -S`, or `llvm-objdump -d`
of a code object) in program order -- prefix, two trips of every loop, suffix
-- with the hardware's rule "loads and stores retire in issue order;
`vmcnt(N)` returns when at most N are outstanding", and reports

  * every instruction that reads or writes a VGPR whose load is in flight,
  * every wait inside a loop and its N (a compiler-inserted `vmcnt(0)` drain
    would show up here),
  * for every load the number of instructions issued before its data is
    waited for (the real prefetch distance).

Usage: waitcheck.py file.s [kernel-name-substring]      (exit status 1 on a
violation); tests/test_codegen.py runs it on the generated kernels.
"""
import re
import subprocess
import sys
from typing import Dict, List, Optional, Set, Tuple

_VREG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
_LABEL = re.compile(r'^([.\w$]+):')
_VMCNT = re.compile(r'vmcnt\((\d+)\)')


def _regs(text: str) -> Set[int]:
  out: Set[int] = set()
  for m in _VREG.finditer(text):
    if m.group(1) is not None:
      out.add(int(m.group(1)))
    else:
      out.update(range(int(m.group(2)), int(m.group(3)) + 1))
  return out


def kernels_of(asm: str) -> Dict[str, List[str]]:
  """{kernel symbol: instruction lines (labels kept)} of a `.s` text."""
  out: Dict[str, List[str]] = {}
  cur: Optional[List[str]] = None
  for raw in asm.splitlines():
    line = raw.split(';')[0].rstrip() if not raw.lstrip().startswith(
        ';;#') else ''
    if not line.strip():
      continue
    m = _LABEL.match(line.strip())
    if m and not line.startswith((' ', '\t')) and not m.group(1).startswith(
        '.L'):
      cur = out.setdefault(m.group(1), [])
      continue
    if cur is None:
      continue
    s = line.strip()
    if s.startswith('.') and not _LABEL.match(s):
      continue                        # directive
    cur.append(s)
    if s.startswith('s_endpgm'):
      cur = None
  return {k: v for k, v in out.items() if any(
      x.startswith('s_endpgm') for x in v)}


def check(lines: List[str]) -> dict:
  """Replays one kernel along every path (conditional branches fork, every
  loop runs two trips); see the module docstring."""
  labels = {}
  for i, s in enumerate(lines):
    m = _LABEL.match(s)
    if m:
      labels[m.group(1)] = i
  in_loop = set()
  nloops = 0
  for i, s in enumerate(lines):
    if s.startswith(('s_cbranch', 's_branch')):
      tgt = s.split()[-1]
      if tgt in labels and labels[tgt] < i:
        in_loop.update(range(labels[tgt], i + 1))
        nloops += 1

  violations: List[str] = []
  seen_violation = set()
  loop_waits: Dict[int, int] = {}
  distances: List[int] = []
  counts = dict(loads=0, stores=0)
  counted = set()
  visited = set()
  # a path: (pc, in-flight FIFO, taken back edges, instructions issued so far)
  work = [(0, (), frozenset(), 0)]
  while work:
    pc, fifo_t, taken, step = work.pop()
    fifo = list(fifo_t)
    while pc < len(lines):
      key = (pc, tuple((k, tuple(sorted(d))) for k, d, _, _ in fifo), taken)
      if key in visited:
        break
      visited.add(key)
      s = lines[pc]
      if _LABEL.match(s):
        pc += 1
        continue
      mnem = s.split()[0]
      ops = s[len(mnem):]
      step += 1
      if mnem == 's_endpgm':
        break
      if mnem in ('s_branch',) or mnem.startswith('s_cbranch'):
        tgt = labels.get(s.split()[-1])
        if tgt is None:
          pc += 1
          continue
        if tgt <= pc:                       # back edge: one extra trip
          if pc not in taken:
            work.append((tgt, tuple(fifo), taken | {pc}, step))
          if mnem == 's_branch':
            break
          pc += 1
          continue
        if mnem == 's_branch':
          pc = tgt
          continue
        work.append((tgt, tuple(fifo), taken, step))   # fork: branch taken
        pc += 1
        continue
      if mnem == 's_waitcnt':
        m = _VMCNT.search(s)
        if m:
          n = int(m.group(1))
          if pc in in_loop:
            loop_waits[pc] = n
          while len(fifo) > n:
            kind, _, at, _ = fifo.pop(0)
            if kind == 'load':
              distances.append(step - at)
        pc += 1
        continue
      touched = _regs(ops)
      busy = set()
      for kind, dest, _, _ in fifo:
        busy |= dest
      if touched & busy and pc not in seen_violation:
        seen_violation.add(pc)
        who = [(at, len(fifo) - 1 - i) for i, (_, dest, _, at) in
               enumerate(fifo) if dest & touched]
        violations.append('line %d: `%s` touches v%s while its load (line %d, '
                          '%d younger in flight) has not been waited for' %
                          (pc, s, sorted(touched & busy), who[0][0], who[0][1]))
      if mnem.startswith(('buffer_load', 'global_load', 'flat_load',
                          'scratch_load')):
        if ' lds' in ops:
          fifo.append(('dma', frozenset(), step, pc))
        else:
          fifo.append(('load', frozenset(_regs(ops.split(',')[0])), step, pc))
          if pc not in counted:
            counted.add(pc)
            counts['loads'] += 1
      elif mnem.startswith(('buffer_store', 'global_store', 'flat_store',
                            'scratch_store', 'buffer_atomic',
                            'global_atomic')):
        fifo.append(('store', frozenset(), step, pc))
        if pc not in counted:
          counted.add(pc)
          counts['stores'] += 1
      pc += 1
  return dict(violations=violations, loop_waits=sorted(loop_waits.values()),
              load_to_wait_instructions=(min(distances) if distances else None,
                                         max(distances) if distances else None),
              loads=counts['loads'], stores=counts['stores'], loops=nloops,
              scratch=any(x.split()[0].startswith('scratch_') for x in lines
                          if not _LABEL.match(x)))


def compile_to_asm(source: str, options) -> str:
  """HIP text -> gfx950 ISA text with the options the JIT uses (needs hipcc,
  no GPU)."""
  import os
  import tempfile
  hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
  with tempfile.TemporaryDirectory() as d:
    src = os.path.join(d, 'k.hip')
    with open(src, 'w') as f:
      f.write(source)
    out = os.path.join(d, 'k.s')
    opts = [o for o in options if not o.startswith('--offload-arch')]
    proc = subprocess.run([hipcc, '--offload-arch=gfx950', '--cuda-device-only',
                           '-S', '-o', out, src] + list(opts),
                          capture_output=True, text=True)
    if proc.returncode != 0:
      raise RuntimeError(proc.stderr)
    with open(out) as f:
      return f.read()


def main():
  with open(sys.argv[1]) as f:
    asm = f.read()
  want = sys.argv[2] if len(sys.argv) > 2 else ''
  bad = 0
  for name, lines in kernels_of(asm).items():
    if want not in name:
      continue
    res = check(lines)
    print(name)
    for k, v in res.items():
      if k != 'violations':
        print('  %-28s %s' % (k, v))
    for v in res['violations'][:20]:
      print('  VIOLATION', v)
    bad += len(res['violations'])
  sys.exit(1 if bad else 0)


if __name__ == '__main__':
  main()
