.LBB8_264:                              ; =>This Inner Loop Header: Depth=1
	scratch_load_dwordx2 v[2:3], off, s8
	s_waitcnt vmcnt(0)
	v_cmp_le_f64_e32 vcc, 0, v[2:3]
	s_and_saveexec_b64 s[6:7], vcc
	s_cbranch_execz .LBB8_263
; %bb.265:                              ;   in Loop: Header=BB8_264 Depth=1
	scratch_load_dwordx2 v[2:3], off, s9
	s_waitcnt vmcnt(0)
	v_cmp_lt_f64_e32 vcc, v[2:3], v[90:91]
	s_nop 1
	v_cndmask_b32_e32 v91, v91, v3, vcc
	v_cndmask_b32_e32 v90, v90, v2, vcc
	s_branch .LBB8_263
.LBB8_266:
	s_or_b64 exec, exec, s[4:5]
.LBB8_267:
	v_writelane_b32 v126, s86, 59
                                        ; implicit-def: $vgpr127 : SGPR spill to VGPR lane
	scratch_store_dword off, v43, off offset:1564 ; 4-byte Folded Spill
	scratch_store_dwordx2 off, v[46:47], off offset:1752 ; 8-byte Folded Spill
	scratch_store_dwordx2 off, v[42:43], off offset:1744 ; 8-byte Folded Spill
	v_writelane_b32 v126, s87, 60
	v_writelane_b32 v126, s84, 61
	s_nop 1
	v_writelane_b32 v126, s85, 62
	v_writelane_b32 v126, s52, 63
	s_nop 1
	v_writelane_b32 v127, s53, 0
	v_writelane_b32 v127, s54, 1
	v_writelane_b32 v127, s55, 2
	v_writelane_b32 v127, s48, 3
	s_nop 1
	v_writelane_b32 v127, s49, 4
	v_writelane_b32 v127, s50, 5
	v_writelane_b32 v127, s51, 6
	s_or_b64 exec, exec, s[2:3]
	s_mov_b32 s2, 0x6dc9c883
	s_mov_b32 s3, 0x3fe45f30
	v_mul_f64 v[0:1], v[44:45], s[2:3]
	s_mov_b32 s2, 0x54400000
