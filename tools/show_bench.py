import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "dice_last", d["train_dice_last_step"])
print("roofline", d["roofline"]["frac"], d["roofline"]["sampled_steps"], "excl", d["roofline_exclusive"]["frac"], "step_mfma", d["step_mfma_frac"])
print("continuity", d.get("continuity"))
print("val_dice", {k:v for k,v in d["val_dice"].items() if k in ("soft","hard_cfg5_volume","patches_per_s_on_this_task","seconds")})
print("reference_api", d["reference_api"])
print("cfg3", d["secondary"]["cfg3"]); print("cfg4", d["secondary"]["cfg4"])
print(d["data"])
