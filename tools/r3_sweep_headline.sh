for ct in 1e-5 3e-5 1e-4; do
CTOL=$ct NOREF=1 MODES="a:f32:1:0:0:4:30:1e-5,b:f32:2:0:0:4:30:1e-5,c:f32:1:0:0:5:40:1e-5,d:f32:1:0:0:4:50:1e-5,e:f32:1:0:0:3:20:1e-5,f:f32:1:0:0:6:60:1e-5" python tools/exp.py 2>&1 | grep -v amdgpu.ids | sed "s/^/ctol $ct /"
done
