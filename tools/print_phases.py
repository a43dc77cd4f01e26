import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f)); t=d.get("time_to_tol",{})
    print(f, "it/s %.0f"%d["value"], "phases", d["phases_s"], "ttt %.3f"%t.get("seconds",0), t.get("iterations"), t.get("phases_s"))
