// TEST INFRASTRUCTURE ONLY -- x86-64 fiber context switch for tests/hipemu.
// void hipemu_switch(void** from_sp, void* to_sp): saves the callee-saved
// registers on the current stack, stores rsp in *from_sp, switches to to_sp.
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch,.-hipemu_switch
)");
