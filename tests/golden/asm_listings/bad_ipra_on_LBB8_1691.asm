	v_mov_b32_e32 v2, -1
	v_readlane_b32 s3, v167, 5
	scratch_store_dword off, v2, off offset:520
	scratch_store_dwordx2 off, v[0:1], off offset:568
	v_mov_b32_e32 v0, 1
	s_andn2_b64 vcc, exec, s[2:3]
	scratch_store_dword off, v0, off offset:512
	s_cbranch_vccnz .LBB8_1691
; %bb.1690:                             ;   in Loop: Header=BB8_269 Depth=1
	scratch_store_byte off, v0, off offset:516
.LBB8_1691:                             ;   in Loop: Header=BB8_269 Depth=1
	s_mov_b32 s50, s71
	s_mov_b32 s71, s70
	s_mov_b64 s[54:55], s[46:47]
	scratch_store_dword off, v26, off offset:1400 ; 4-byte Folded Spill
	scratch_store_dwordx2 off, v[70:71], off offset:1352 ; 8-byte Folded Spill
	scratch_store_dwordx2 off, v[68:69], off offset:1344 ; 8-byte Folded Spill
	s_or_b64 exec, exec, s[4:5]
	s_and_saveexec_b64 s[64:65], s[0:1]
	s_cbranch_execz .LBB8_1796
; %bb.1692:                             ;   in Loop: Header=BB8_269 Depth=1
	s_mov_b64 s[0:1], 0
	s_cmp_lt_i32 s69, 2
	s_mov_b64 s[100:101], exec
	s_mov_b64 exec, -1
	scratch_load_dword v166, off, off offset:1336 ; 4-byte Folded Reload
