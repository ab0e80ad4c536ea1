import sys, torch
sys.path.insert(0, '.')
from nero_amd.synthetic import synthetic_rays
from nero_amd import _lib as L
from nero_amd import stage1
o, d, _, _ = synthetic_rays(200000, seed=3)
o, d = o.cuda(), d.cuda()
a = torch.sum(d ** 2, dim=-1, keepdim=True)
b = 2.0 * torch.sum(o * d, dim=-1, keepdim=True)
mid = 0.5 * (-b) / a
x, y, z = (d[:, i:i + 1] * d[:, i:i + 1] for i in range(3))
print('a == (x+y)+z', bool((a == (x + y) + z).all()), ' a == x+(y+z)', bool((a == x + (y + z)).all()), ' a == (x+z)+y', bool((a == (x + z) + y).all()))
p, q, r = (o[:, i:i + 1] * d[:, i:i + 1] for i in range(3))
s1 = (p + q) + r
print('sum(o*d) == (p+q)+r', bool((torch.sum(o * d, -1, keepdim=True) == s1).all()), ' p+(q+r)', bool((torch.sum(o * d, -1, keepdim=True) == p + (q + r)).all()))
n2, f2 = torch.empty_like(a), torch.empty_like(a)
R = o.shape[0]
L.check(stage1._lib.nero_near_far_sphere(o.data_ptr(), d.data_ptr(), R, n2.data_ptr(), f2.data_ptr(), L.stream_ptr()))
far = mid + 1.0
print('kernel far == torch far', int((f2 != far).sum()), 'of', R)
# double-precision emulation of variants
ad, bd = a.double(), b.double()
for name, m in (('0.5*(-b)/a in fp32 via fp64 div', ((0.5 * (-b)).double() / ad).float()), ('(-b)*(0.5/a)', (-b) * (0.5 / a)), ('(0.5*(-b)) * (1/a)', (0.5 * (-b)) * (1.0 / a))):
    print(name, 'mismatch vs torch mid:', int((m != mid).sum()), ' vs kernel far-1:', int(((m + 1.0) != f2).sum()))
