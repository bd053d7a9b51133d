import faulthandler, os, sys
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch  # noqa
import server_graph as sg, fused_graph as fg
PKG = os.path.join(fg.ROOT, "mediastreamer2_amd")
h = fg.Host(PKG)
name = sys.argv[1]
r = sg.run(PKG, sys.argv[2] == "fused", sg.SCENARIOS[name], h)
print(name, sys.argv[2], r["stats"], r["late"], [len(x) for x in r["out"]])
