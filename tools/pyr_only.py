import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daala_amd as D
D.init(0)
F=8
luma=torch.randint(0,256,(F,1088,1920),dtype=torch.uint8,device='cuda')
lv=D.forward_pyramid(luma,0,1920,1080)
for _ in range(5): D.forward_pyramid(luma,0,1920,1080,levels=lv)
torch.cuda.synchronize()
