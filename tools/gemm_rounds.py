import sys, os
sys.path.insert(0, os.getcwd())
sys.argv = [sys.argv[0], "none"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench_ops", "tools/bench_ops.py")
b = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(b)
except SystemExit:
    pass
for m in (16384, 32768, 65536):
    b.linear(m, 1152, 1152)
    b.linear(m, 1152, 1152, out_f32=1, res=2)
    b.linear(m, 4608, 1152, out_f32=1, res=2)
    b.linear(m, 1152, 4608, act=2)
