import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for n in a:
    same = torch.equal(a[n][0], b[n][0]) and torch.equal(a[n][1], b[n][1]) and all(torch.equal(a[n][2][k], b[n][2][k]) for k in a[n][2])
    print(f"n {n}: bit-identical probabilities (2 forwards) and running statistics: {same}; finite {bool(torch.isfinite(a[n][0]).all())}")
