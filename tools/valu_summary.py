"""profiles/<tag>_valu.json from a rocprofv3 --pmc pass with the SQ counters below (dev tool).
    python tools/valu_summary.py TAG PMC_DIR "<command that was profiled>" """
import glob, json, sqlite3, sys
tag, path, command = sys.argv[1], sys.argv[2], sys.argv[3]
db = sqlite3.connect(glob.glob(path + "/**/*.db", recursive=True)[0])
rows = db.execute("select counter_name, avg(value), count(*), avg(duration) from counters_collection "
                  "where kernel_name like '%k_llcs_diag%' group by counter_name").fetchall()
c = {name: value for name, value, _, _ in rows}
n, dur = rows[0][2], rows[0][3]
waves = c["SQ_WAVES"]
derived = {
    "valu_instructions_per_wave": c["SQ_INSTS_VALU"] / waves,
    "wave_lifetime_cycles": 4 * c["SQ_WAVE_CYCLES"] / waves,
    "valu_active_cycles_per_wave": 4 * c["SQ_ACTIVE_INST_VALU"] / waves,
    "valu_active_fraction_of_wave_lifetime": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
    "issue_stall_fraction_of_wave_lifetime": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
    "lane_ops_per_s_executed": 64 * c["SQ_INSTS_VALU"] / (dur * 1e-9),
    "executed_fraction_of_78.6T_peak": 64 * c["SQ_INSTS_VALU"] / (dur * 1e-9) / 78.6e12,
}
out = {"command": command, "kernel": "k_llcs_diag<3,2,true>", "launches_sampled": n, "avg_duration_ns_under_pmc": dur,
       "per_launch": c, "units": "SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md); "
       "GRBM_GUI_ACTIVE is summed over the 8 XCDs", "derived": derived}
json.dump(out, open(f"profiles/{tag}_valu.json", "w"), indent=1)
print(json.dumps(out, indent=1))
