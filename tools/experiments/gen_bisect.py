import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import fuzz
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower
text,dim,_=fuzz.program(613)
st=core.from_text(text)
opts = runtime.resolve_options(st, lower.LowerOptions(fuse=(2,), peel=-1), (1100,207))
src = lower.lower(st, opts).source
for L in [int(a) for a in sys.argv[1:]]:
    runtime.compile_source(src, '%s.hip' % st.app_name, options=runtime.COMPILE_OPTIONS+('-mllvm','-opt-bisect-limit=%d' % L))
print('ok')
