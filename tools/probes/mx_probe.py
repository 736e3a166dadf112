import ctypes as C, subprocess, sys, torch
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "tools/probes/mx_probe.hip", "-o", "/tmp/mx_probe.so"])
lib = C.CDLL("/tmp/mx_probe.so")
p = lambda t: C.c_void_p(t.data_ptr())
D = torch.zeros(16, 16, device="cuda")
def run(A, B, sa=127, sb=127):
    a8, b8 = A.to(torch.float8_e4m3fn).cuda().view(torch.uint8).contiguous(), B.to(torch.float8_e4m3fn).cuda().view(torch.uint8).contiguous()
    lib.mx_run(p(a8), p(b8), p(D), sa, sb); torch.cuda.synchronize()
    return D.cpu().clone()
for (i0, k0, j1, k1) in ((3, 5, 7, 5), (3, 5, 7, 6), (3, 40, 7, 40), (3, 40, 7, 8), (0, 0, 0, 0), (3, 100, 7, 100)):
    A = torch.zeros(16, 128); B = torch.zeros(16, 128); A[i0, k0] = 1.0; B[j1, k1] = 2.0
    d = run(A, B)
    nz = [(int(r), int(c), float(d[r, c])) for r, c in d.nonzero()]
    print(f"A[{i0}][{k0}] = 1, B[{j1}][{k1}] = 2 -> nonzero D entries:", nz)
g = torch.Generator().manual_seed(0)
A = (torch.randn(16, 128, generator=g) * 2).to(torch.float8_e4m3fn).float(); B = (torch.randn(16, 128, generator=g) * 2).to(torch.float8_e4m3fn).float()
ref = A @ B.T
d = run(A, B)
print("random: max |D - ref|", (d - ref).abs().max().item(), " max |D - ref^T|", (d - ref.T).abs().max().item(), " |ref| max", ref.abs().max().item())
# per-lane block scales: D[i][j] = sum_blocks 2^(SA[i][blk] - 127 + SB[j][blk] - 127) * dot(A[i][blk], B[j][blk])
SA = torch.randint(120, 134, (16, 4), generator=g, dtype=torch.uint8); SB = torch.randint(120, 134, (16, 4), generator=g, dtype=torch.uint8)
Ab = A.view(16, 4, 32) * torch.exp2(SA.float() - 127)[..., None]; Bb = B.view(16, 4, 32) * torch.exp2(SB.float() - 127)[..., None]
ref = Ab.reshape(16, 128) @ Bb.reshape(16, 128).T
a8, b8 = A.to(torch.float8_e4m3fn).cuda().view(torch.uint8).contiguous(), B.to(torch.float8_e4m3fn).cuda().view(torch.uint8).contiguous()
sa, sb = SA.cuda(), SB.cuda()
for sel in range(4):
    lib.mx_scale_run(p(a8), p(b8), p(sa), p(sb), p(D), sel); torch.cuda.synchronize()
    d = D.cpu()
    print(f"per-lane block scales, byte select {sel}: max |D - ref| {(d - ref).abs().max().item():.4g}  (|ref| max {ref.abs().max().item():.4g})")
# which (row, K block) does the scale byte 0 of lane L apply to?  A = ones in one K block, B = ones, one lane's A scale = 128
ones8 = torch.ones(16, 128).to(torch.float8_e4m3fn).cuda().view(torch.uint8).contiguous()
unit = torch.full((64,), 127, dtype=torch.int32).cuda()
amap, bmap = {}, {}
for blk in range(4):
    Ak = torch.zeros(16, 128); Ak[:, 32 * blk:32 * blk + 32] = 1.0
    ak8 = Ak.to(torch.float8_e4m3fn).cuda().view(torch.uint8).contiguous()
    for L in range(64):
        sw = unit.clone(); sw[L] = 128
        lib.mx_scale_raw_run(p(ak8), p(ones8), p(sw), p(unit), p(D)); torch.cuda.synchronize()
        d = D.cpu()
        rows = [int(i) for i in range(16) if float(d[i, 0]) == 64.0]
        if rows: amap.setdefault(L, []).append((rows, blk))
        lib.mx_scale_raw_run(p(ones8), p(ak8), p(unit), p(sw), p(D)); torch.cuda.synchronize()
        d = D.cpu()
        cols = [int(j) for j in range(16) if float(d[0, j]) == 64.0]
        if cols: bmap.setdefault(L, []).append((cols, blk))
print("A-operand scale of lane L applies to (rows, K block):", {L: amap.get(L) for L in (0, 1, 15, 16, 17, 32, 48, 63)})
print("B-operand scale of lane L applies to (cols, K block):", {L: bmap.get(L) for L in (0, 1, 15, 16, 17, 32, 48, 63)})
ok_a = all(amap.get(L) == [([L & 15], L >> 4)] for L in range(64)); ok_b = all(bmap.get(L) == [([L & 15], L >> 4)] for L in range(64))
print("lane (fi, fg) scales row fi, block fg:", ok_a, ok_b)
Ak = torch.zeros(16, 128); Ak[:, 0:32] = 1.0
ak8 = Ak.to(torch.float8_e4m3fn).cuda().view(torch.uint8).contiguous()
for L in (0, 5, 16, 37):
    sw = unit.clone(); sw[L] = 128
    lib.mx_scale_raw_run(p(ak8), p(ones8), p(sw), p(unit), p(D)); torch.cuda.synchronize()
    d = D.cpu()
    print("A block 0 = ones, lane", L, "A scale 128: distinct D values", sorted(set(d.flatten().tolist())), "positions != 32:", [(int(r), int(c)) for r, c in (d != 32).nonzero()][:20])
sw = torch.full((64,), 130, dtype=torch.int32).cuda()
lib.mx_scale_raw_run(p(ak8), p(ones8), p(sw), p(unit), p(D)); torch.cuda.synchronize(); print("all lanes A scale 130:", sorted(set(D.cpu().flatten().tolist())))
lib.mx_scale_raw_run(p(ak8), p(ones8), p(unit), p(sw), p(D)); torch.cuda.synchronize(); print("all lanes B scale 130:", sorted(set(D.cpu().flatten().tolist())))
