import sys, time, os, tempfile
sys.path.insert(0, os.getcwd())
from maniac_mc_amd import synth, io_maniac, run
s = synth.spce_box(15)
d = tempfile.mkdtemp()
files = io_maniac.write_input_files(s, d + "/in", nb_block=2, nb_step=15000, translation_step=0.3, rotation_step_angle=0.3,
                                    translation_proba=0.5, rotation_proba=0.5, masses=[15.9994, 1.008], atom_names=["OW", "HW"])
t0 = time.perf_counter()
res = run.run_simulation(*files, d + "/out/", seed=5)
el = time.perf_counter() - t0
print("chain: 30000 moves in %.2f s total (incl. setup, file output) -> %.0f moves/s; accepted %d" % (el, 30000 / el, res["counters"][1] + res["counters"][3]))
