import sys; sys.path.insert(0,'/root/repo')
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
for w in ["ecadd0w","ecadd1w","ecadd2w","ecadd17w","ecadd65w","ecadd0f","ecadd1f","ecadd17f"]:
    print(w, [round(B.ubench(w),1) for _ in range(2)])
