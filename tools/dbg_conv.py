import sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
from conftest import seeded_rand
from opticalflow_amd import ops
dev = torch.device('cuda:0')
for (B, cin, cout, H, W) in [(1,565,128,16,32), (1,560,128,16,32), (1,568,128,16,32), (1,565,128,8,32), (1,200,128,16,32), (1, 565, 32, 16, 32)]:
    x = seeded_rand((B, cin, H, W), 60, -1, 1)
    w = seeded_rand((cout, cin, 3, 3), 61, -1, 1) * (2.0 / (cin * 9)) ** 0.5
    bias = seeded_rand((cout,), 62, -0.5, 0.5)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), bias.double(), padding=1), 0.1)
    wp = ops.pack_conv3x3(w.to(dev))
    for rep in range(2):
        got = ops.conv3x3(x.to(dev), wp, bias.to(dev), cout).cpu()
        err = (got.double() - ref).abs()
        print((B,cin,cout,H,W), 'rep', rep, 'max err', err.max().item(), 'bad frac', (err > 1e-3).float().mean().item())
    if err.max() > 1e-3:
        bad = (err > 1e-3)
        print('  bad per cout-tile:', [bad[0, i*32:(i+1)*32].float().mean().item() for i in range((cout+31)//32)])
        print('  bad per row:', [round(bad[0, :, r].float().mean().item(),3) for r in range(H)])
        print('  bad per col:', [round(bad[0, :, :, c].float().mean().item(),3) for c in range(W)])
