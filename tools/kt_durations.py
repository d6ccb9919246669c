import csv,glob,sys
for k in sys.argv[1:]:
    for f in glob.glob(f"gpurun_out/to_{k}/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "k_trace" in r["Kernel_Name"]:
                n=int(r["Grid_Size_X"]); us=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
                print(k, n, f"{us:.1f} us  {n/us/1e3:.2f} Grays/s")
