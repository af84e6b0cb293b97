"""Doc tooling: fills the 'Measured (MI355X, round 3)' section of DESIGN.md from profiles/r3_1gb_bench.json and the rocprof kernel summary
(template: tools/dbg/r3_measured.md.in).  usage: python tools/dbg/fill_measured.py"""
import json, csv, os, re
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d=json.loads(open(R+'/profiles/r3_1gb_bench.json').read().strip().splitlines()[-1])
c=d["cpu_baseline"]; pb=c["python_boundary"]; x=d["extra"]
front=sum(d["kernels"][k]["ms_total"] for k in ("char_hist","segments","dedup","build","pair_count"))
sp=lambda v:"{:,.0f}".format(v).replace(","," ")
rows={}
for line in open(R+'/profiles/r3_1gb_kernel_stats.csv').read().splitlines()[1:]:
    m=re.match(r"(.*),(\d+),([\d.]+),([\d.]+),([\d.]+),([\d.]+),([\d.]+)$", line)
    if m: rows[m.group(1)]=(int(m.group(2)), float(m.group(3)))
kw=sum(v[1] for k,v in rows.items() if k.startswith("k_words<"))
kwf=sum(v[1] for k,v in rows.items() if k.startswith("k_words<") and k.rstrip().endswith("true>"))
kwfn=sum(v[0] for k,v in rows.items() if k.startswith("k_words<") and k.rstrip().endswith("true>"))
kt=sum(v[1] for k,v in rows.items() if k.startswith("k_tiles<512, 8, true"))
idx=sum(v[1] for k,v in rows.items() if k.startswith("k_idx_stream"))
f={
"ABCD_S":"%.4f"%(d["ms_per_step"]/1e3),"ABCD_MBS":sp(d["value"]),"ABCD_MS":"%.1f"%d["ms_per_step"],
"E2E_S":"%.4f"%d["e2e"]["train_file_to_model"]["seconds"],"E2E_MBS":sp(d["value_file_to_model"]),
"ZIPF_S":"%.4f"%(x["zipf"]["ms_per_step"]/1e3),"ZIPF_MBS":sp(x["zipf"]["value"]),"ZIPF_US":"%.1f"%x["zipf"]["us_per_round"],
"CJK_S":"%.3f"%(x["cjk"]["ms_per_step"]/1e3),"CJK_MBS":sp(x["cjk"]["value"]),
"Z4_S":"%.4f"%(x["zipf4m"]["ms_per_step"]/1e3),"Z4_MBS":sp(x["zipf4m"]["value"]),
"ENC":"%.2f·10⁸"%(d["encode"]["value"]/1e8),"ENC_MS":"%.1f"%d["encode"]["kernel_ms"],"ENC_NC_MS":"%.1f"%d["encode"]["word_cache"]["without_cache_kernel_ms"],
"DROP":"%.2f·10⁷"%(d["encode_dropout"]["value"]/1e7),"DROP_MS":"%.1f"%d["encode_dropout"]["kernel_ms"],"KS":"%.4f"%d["encode_dropout"]["distribution"]["ks_sentence_lengths"],
"KSC":"%.4f"%d["encode_dropout"]["distribution"]["ks_critical_alpha_0.001"],"CHI":"%.2f"%d["encode_dropout"]["distribution"]["chi2_per_dof_unigram_ids"],
"PYAPI":"%.1f·10⁶"%(pb["encode_list_drop_in"]["value"]/1e6),"PYREF":"%.2f·10⁵"%(pb["encode_list_reference"]["value"]/1e5),
"CLI":"%.2f·10⁶"%(pb["cli_encode_drop_in"]["value"]/1e6),"CLIREF":"%.2f·10⁵"%(pb["cli_encode_reference"]["value"]/1e5),
"CPU":"%.2f"%c["value"],"CPU_S":"%.1f"%c["train_seconds"],"GPUCPU":"%.0f"%c["gpu_over_cpu"],"E2ECPU":"%.0f"%c["gpu_e2e_over_cpu"],
"FRONT":"%.1f"%front,"K2A":"%.2f"%d["kernels"]["segments"]["ms_total"],"K4":"%.1f"%d["kernels"]["merge_apply"]["ms_total"],"CAND":"%.1f"%d["kernels"]["cand_scan"]["ms_total"],"CANDN":str(d["kernels"]["cand_scan"]["launches"]),
"REST":"%.0f"%(d["ms_per_step"]-front-d["kernels"]["merge_apply"]["ms_total"]-d["kernels"]["cand_scan"]["ms_total"]),
"ACH":"%.0f"%d["roofline"]["achieved"],"FRAC":"%.2f"%(100*d["roofline"]["frac"]),"TRAF":"%.0f"%(d["roofline"]["traffic"]/1e6),"TOA":"%.1f"%d["roofline"]["traffic_over_algorithmic"],
"KWF":"%.1f"%kwf,"KWFN":str(kwfn),"KWU":"%.1f"%(kw-kwf),"KDA":"%.1f"%rows["k_delta_apply"][1],"KDAN":str(rows["k_delta_apply"][0]),"KG":"%.1f"%rows["k_wgather"][1],"KT":"%.1f"%kt,"IDX":"%.1f"%idx,
}
s=open(R+'/tools/dbg/r3_measured.md.in').read()
for k,v in f.items(): s=s.replace("{%s}"%k, v)
left=re.findall(r"\{[A-Z0-9_]+\}", s)
assert not left, left
p=R+'/DESIGN.md'; t=open(p).read()
a=t.index("### Measured (MI355X, round 3;"); b=t.index("### Measured (MI355X, round 2;")
t=t[:a]+s.rstrip("\n")+"\n\n"+t[b:]
open(p,'w').write(t)
print("ok", f["ABCD_MS"], f["K4"], f["CAND"], f["FRONT"], f["IDX"], "hip_events", d["roofline"].get("avg_launch_ms_hip_events"))
