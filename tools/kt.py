import json,sys
r=json.loads(sys.stdin.read())
k=r["roofline"]["kernels"]
print(r["ms_per_step"], {n:(v["ms"],v["calls"]) for n,v in k.items() if n.startswith("block_kernel")})
