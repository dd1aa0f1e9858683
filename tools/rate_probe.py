import sys; sys.path.insert(0,'/root/repo')
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
for w in ["modmul","modmul_o4","modmul_o2","modmul_o1","modmul29","modmul29_o4","modmul29_o3","modmul29_o2","modmul29_o1","modmul29_check"]: print(w, "%.3e"%B.ubench(w))
