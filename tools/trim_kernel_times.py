"""Per-launch durations of k_trim from a rocprofv3 --kernel-trace directory:   python tools/trim_kernel_times.py gpurun_out/trim"""
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_trim' in r['Kernel_Name']:
        print("%-12s %9.3f ms  grid %8s  scratch %s B/lane" % (r['Kernel_Name'].split('(')[0][5:], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r['Grid_Size_X'], r['Scratch_Size']))
