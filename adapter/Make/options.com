/* OpenFOAM.com (the reference's src/Make/options.com) + libsmgpu.  Copied to Make/options by adapter/Allwmake, which also sets
   VERSION_SPECIFIC_INC (-DOPENFOAM_COM [-DSMGPU_WITH_RCCL]), SMGPU_ROOT, ROCM_PATH and SMGPU_COMM_LIBS (-lrccl). */
EXE_INC = \
    $(VERSION_SPECIFIC_INC) \
    -I$(LIB_SRC)/finiteVolume/lnInclude \
    -I$(LIB_SRC)/meshTools/lnInclude \
    -I$(LIB_SRC)/surfMesh/lnInclude \
    -I$(SMGPU_ROOT)/include \
    -I$(ROCM_PATH)/include \
    -D__HIP_PLATFORM_AMD__

EXE_LIBS = \
    -lfiniteVolume \
    -lmeshTools \
    -lsurfMesh \
    -L$(SMGPU_ROOT)/smoothmesh_amd/csrc -lsmgpu \
    -L$(ROCM_PATH)/lib -lamdhip64 $(SMGPU_COMM_LIBS) \
    -Wl,-rpath,$(SMGPU_ROOT)/smoothmesh_amd/csrc
