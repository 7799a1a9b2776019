import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'][:34], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'gemm_bf16' in r['Kernel_Name']]
i = 0
allshapes = [('cube 8192', 8192, 8192, 8192), ('cube 4096', 4096,4096,4096), ('logits', 132096, 8192, 512), ('enc gi L2', 131072, 3072, 1024), ('dec gi', 132096, 1536, 512), ('dho', 132096, 512, 8192), ('enc dX', 131072, 1024, 3072), ('dec dX', 132096, 512, 1536)]
pre = sys.argv[2] if len(sys.argv) > 2 else ''
for n, M, N, K in [x for x in allshapes if x[0].startswith(pre)]:
    a = seq[i:i+7]; b = seq[i+7:i+14]; i += 14
    ta = sum(x[1] for x in a[2:]) / 5; tb = sum(x[1] for x in b[2:]) / 5
    fl = 2.0 * M * N * K
    print('%-10s nt256 %8.1f us %6.0f TF | p8 %8.1f us %6.0f TF | C write %.2f GB -> %.2f TB/s' % (n, ta, fl / ta / 1e6, tb, fl / tb / 1e6, M * N * 4 / 1e9, M * N * 4 / tb / 1e6))
for n, M, N, K in [x for x in [('tn dE', 8192, 512, 132096), ('tn enc dW', 3072, 1024, 131072), ('tn dec dW', 1536, 512, 132096)] if x[0].startswith(pre)]:
    a = seq[i:i+7]; b = seq[i+7:i+14]; i += 14
    if len(b) < 7: break
    ta = sum(x[1] for x in a[2:]) / 5; tb = sum(x[1] for x in b[2:]) / 5
    fl = 2.0 * M * N * K
    print('%-10s tn256 %8.1f us %6.0f TF (%s) | p8 %8.1f us %6.0f TF (%s)' % (n, ta, fl / ta / 1e6, a[0][0], tb, fl / tb / 1e6, b[0][0]))
