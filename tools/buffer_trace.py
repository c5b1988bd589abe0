"""GPU-box tool: one line per buffer of cfg2.ttml through the element (FLUTL_TIMING=2):
record / device microseconds and the region cache counters.  python tools/buffer_trace.py"""
import sys,os,subprocess
sys.path.insert(0,'tools')
import gstenv
env=gstenv.env_for(gstenv.HIP_PLUGIN_DIR, {"FLUTL_TIMING":"2"})
cmd=[os.path.join(gstenv.CONDA,"bin","gst-launch-1.0"),"-q","filesrc","location="+os.path.abspath("bench_data/cfg2.ttml"),"!","ttmlrender","!","video/x-raw,width=3840,height=2160","!","fakesink"]
for i in range(2):
    r=subprocess.run(cmd,env=env,capture_output=True,text=True,cwd="/tmp",timeout=120)
print(r.stderr[-30000:])
