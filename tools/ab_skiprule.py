"""Development aid: the tandem-repeat skip-rule case of tests/test_engine_emu.py against a given build of the library
(T4_LIB=... python tools/ab_skiprule.py), with the first differences printed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import trust4_amd
import test_engine_emu as te
print("library:", trust4_amd.lib_path())
eng = trust4_amd.Engine(0)
for args in ((9, 170), (10, 300), (11, 120)):
    try:
        te.check_novel_min_statistics(eng, seed=args[0], n_contigs=args[1], repeats=True)
        print("seed %d contigs %d: ok" % args)
    except AssertionError as e:
        print("seed %d contigs %d: FAIL %s" % (args[0], args[1], str(e)[:300]))
