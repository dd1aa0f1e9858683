import sys; sys.path.insert(0,'/root/repo')
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
for w in ["ecadd0w","ecadd1w","ecadd17w","ecadd65w","ecadd1q","ecadd17q","ecadd1h","ecadd17h","ecadd65h","ecaddtw","ecaddth","ecadduw","ecadduh","ecaddvw","ecaddvh","ecaddxw","ecaddxh","ecadd6w","ecadd6h","ecadd0f","ecadd1f","ecadd17f"]:
    print(w, [round(B.ubench(w),1) for _ in range(2)])
