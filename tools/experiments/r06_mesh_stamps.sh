cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PDFNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/probe/libpdfnet_hip_mdstamps.so timeout 300 python tools/probe/mesh_stamps.py 32 2>&1 | grep -v "^W\|amdgpu.ids" > gpurun_out/r06_mesh_stamps.txt
cat gpurun_out/r06_mesh_stamps.txt
