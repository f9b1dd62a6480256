import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); 
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
t = importlib.import_module('tests.test_k1_fp32_gpu')
for b in (int(a) for a in sys.argv[1:]):
    try:
        rep = t._resnet50_three_steps(b, 1.0)
        print('BATCH', b); print('\n'.join(rep), flush=True)
    except Exception as e:
        print('BATCH', b, 'failed', repr(e)[:500], flush=True)
